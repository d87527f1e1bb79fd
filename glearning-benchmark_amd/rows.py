"""Host-side view of one tokenized epoch: what the per-item call sites of the reference read from.

trainer/train_agtt.py:246-273 fetches one graph at a time (`tokens = self.tokenizer(data)`, 1-D LongTensor).  The
kernels tokenize a whole split per launch, so an epoch crosses to the host ONCE, in the packed form (include/gtok.h,
"packed rows": no padding, rows back to back) and widened to int64 on the device - one pinned D2H copy - and items are
zero-copy slices of that buffer.  A new epoch gets a new buffer: rows handed out earlier stay valid.
"""
import os
import weakref
from typing import Optional, Tuple

import torch

from . import ops as _ops


class EpochRows:
    """Rows of one [G, ld] slab + lengths (device) as slices of one host int64 buffer."""

    def __init__(self, ids: torch.Tensor, lens: torch.Tensor, epoch: int = 0, pin: bool = True, align: int = 8):
        """ids: the int32 slab, or the 16-bit slab of ops.sent(..., u16=True) (int16 storage) - then the rows are packed
        straight to int64 (gtok_pack_rows_u16, elem_bytes 8: no int32 detour, no widening pass).
        pin=False: the buffer is ordinary pageable memory (rows that DataLoader worker processes will read: pinned
        allocations are not reliably inherited across fork).  align=1: rows back to back without the 8-id alignment
        of the packed format's fast path - `all_rows()` can then cut every row in one call."""
        ld = int(ids.shape[1])
        self.align = align
        ptr = _ops.row_offsets(lens, ld, align)
        if ids.dtype == torch.int16:
            wide = _ops.pack_rows_u16(ids, lens, ptr, elem_bytes=8, check_status=False)[0]
        else:
            wide = _ops.pack_rows(ids, lens, ptr, elem_bytes=4, check_status=False)[0].to(torch.int64)
        n = torch.clamp(lens, 0, ld)
        self.epoch = epoch
        self.tokens = torch.empty(wide.shape, dtype=torch.int64, pin_memory=pin)
        self.tokens.copy_(wide, non_blocking=pin)
        ptr_h = torch.empty(ptr.shape, dtype=torch.int64, pin_memory=True); ptr_h.copy_(ptr, non_blocking=True)
        n_h = torch.empty(n.shape, dtype=n.dtype, pin_memory=True); n_h.copy_(n, non_blocking=True)
        torch.cuda.current_stream(ids.device).synchronize()
        self.start = ptr_h[:-1].tolist()            # Python ints: indexing a list is cheaper than a tensor element
        self.count = n_h.tolist()
        self.served = bytearray(len(self.count))

    def __len__(self) -> int:
        return len(self.count)

    def row(self, i: int) -> torch.Tensor:
        s = self.start[i]
        return self.tokens[s:s + self.count[i]]

    def all_rows(self):
        """Every row as a view of the buffer, as a list: one split call for tightly packed rows (what a Python loop of
        slices costs 1.5 us per row for), the loop otherwise."""
        if self.align == 1:
            return list(self.tokens[:sum(self.count)].split(self.count)) if self.count else []
        return [self.row(i) for i in range(len(self.count))]

    def take(self, i: int) -> Optional[torch.Tensor]:
        """Row i, once per epoch: None when it was handed out before (the caller starts a new epoch - a second fetch
        of an item means a new random trail in the reference, which tokenizes on every fetch)."""
        if self.served[i]:
            return None
        self.served[i] = 1
        return self.row(i)


# ---- which split does an item come from? ------------------------------------------------------------------------------
# The reference's per-item loop is `data = self.pyg_dataset[idx]; tokens = self.tokenizer(data)` (train_agtt.py:247-250):
# the tokenizer sees one graph and nothing else.  This package's dataset classes mark what they return with (dataset,
# index), so that the tokenizer can tokenize the item's whole split in one launch and serve rows from it.
_OWNERS = weakref.WeakValueDictionary()      # dataset token -> dataset, for as long as the dataset lives
_NEXT_TOKEN = [1]              # tokens are handed out once and never reused (id() of a freed dataset can come back)
_LAST = [0, -1, None]          # (dataset token, index, weak reference to the object returned): fallback for objects that refuse attributes


def _token_of(owner) -> int:
    """The dataset's key in the registry: a counter value stored on the dataset itself the first time it is asked for."""
    tok = getattr(owner, "_gtok_token", None)
    if tok is None or _OWNERS.get(tok) is not owner:
        tok = _NEXT_TOKEN[0]
        _NEXT_TOKEN[0] += 1
        try:
            owner._gtok_token = tok
        except Exception:
            pass
        _OWNERS[tok] = owner
    return tok


def tag_item(owner, idx: int, data):
    """Mark `data` as item `idx` of `owner`.  The mark is three plain ints - the dataset's token in a weak registry (a
    counter, never reused: an item that outlives its dataset cannot resolve to a later one), the index and the pid that
    made it (a forked DataLoader worker inherits the registry, but a mark from another process is not honoured) - it
    pickles with the object and dies with the dataset."""
    key = _token_of(owner)
    try:
        data._gtok_src = (key, int(idx), os.getpid())  # a private attribute: torch_geometric's Data keeps those out of its keys
    except Exception:
        pass
    try:
        ref = weakref.ref(data)
    except TypeError:                                   # (no __weakref__ slot: the one object that stays referenced is the last one served)
        ref = (lambda obj: (lambda: obj))(data)
    _LAST[0], _LAST[1], _LAST[2] = key, int(idx), ref
    return data


def item_source(data) -> Optional[Tuple[object, int]]:
    """(dataset, index) the item was fetched from, or None (an object built by the caller, or one whose dataset is gone or
    lives in another process: tokenize it on its own)."""
    try:
        src = getattr(data, "_gtok_src", None)
    except Exception:
        src = None
    if src is None and _LAST[2] is not None and _LAST[2]() is data:
        src = (_LAST[0], _LAST[1], os.getpid())
    if not (isinstance(src, tuple) and len(src) == 3) or src[2] != os.getpid():
        return None
    owner = _OWNERS.get(src[0])
    if owner is None or getattr(owner, "_gtok_token", src[0]) != src[0]:
        return None
    return owner, src[1]
