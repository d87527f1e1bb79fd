"""Fixed ZINC vocabulary shared by IBTT and AGTT — same names and ids as the reference's
graph_data_loader/zinc_vocab.py (ids 0-7 specials, 8-16 atoms, 17-20 bonds, 21 'regression',
22+ dynamic).  The id ranges are also compiled into the kernels (gtok_common.hpp:remap_zinc_token)."""
from typing import Dict, Iterable, Tuple

SPECIAL_TOKENS = ["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>"]
ZINC_ATOM_TYPES = ["C", "N", "O", "F", "P", "S", "Cl", "Br", "I"]
ZINC_BOND_TYPES = ["single", "double", "triple", "aromatic"]
NUM_ATOM_TYPES = len(ZINC_ATOM_TYPES)
NUM_BOND_TYPES = len(ZINC_BOND_TYPES)

ATOM_ID_BASE = len(SPECIAL_TOKENS)                  # 8
BOND_ID_BASE = ATOM_ID_BASE + NUM_ATOM_TYPES        # 17
TASK_ID = BOND_ID_BASE + NUM_BOND_TYPES             # 21 ('regression')
DYNAMIC_ID_BASE = TASK_ID + 1                       # 22


def build_fixed_zinc_vocab() -> Tuple[Dict[str, int], Dict[int, str]]:
    """(token -> id, id -> token) for the 22 fixed entries (reference zinc_vocab.py:35-78)."""
    order = SPECIAL_TOKENS + ZINC_ATOM_TYPES + ZINC_BOND_TYPES + ["regression"]
    vocab = {tok: i for i, tok in enumerate(order)}
    return vocab, {i: tok for tok, i in vocab.items()}


def get_atom_type_id(atom_type_idx: int) -> int:
    """PyG atom index 0..8 -> 8..16; ValueError outside (reference :81-96)."""
    if 0 <= atom_type_idx < NUM_ATOM_TYPES:
        return ATOM_ID_BASE + atom_type_idx
    raise ValueError(f"Invalid atom type index: {atom_type_idx} (expected 0-{NUM_ATOM_TYPES - 1})")


def get_bond_type_id(bond_type_idx: int) -> int:
    """1-based bond index 1..4 -> 17..20; ValueError outside (reference :99-115)."""
    if 1 <= bond_type_idx <= NUM_BOND_TYPES:
        return BOND_ID_BASE + bond_type_idx - 1
    raise ValueError(f"Invalid bond type index: {bond_type_idx} (expected 1-{NUM_BOND_TYPES})")


def get_atom_type_from_id(token_id: int) -> str:
    if ATOM_ID_BASE <= token_id < BOND_ID_BASE:
        return ZINC_ATOM_TYPES[token_id - ATOM_ID_BASE]
    raise ValueError(f"Invalid atom type token ID: {token_id} (expected {ATOM_ID_BASE}-{BOND_ID_BASE - 1})")


def get_bond_type_from_id(token_id: int) -> str:
    if BOND_ID_BASE <= token_id < TASK_ID:
        return ZINC_BOND_TYPES[token_id - BOND_ID_BASE]
    raise ValueError(f"Invalid bond type token ID: {token_id} (expected {BOND_ID_BASE}-{TASK_ID - 1})")


def extend_vocab_with_dynamic_tokens(base_vocab: Dict[str, int], dynamic_tokens: Iterable[str]) -> Dict[str, int]:
    """Append unseen tokens in the order given, ids continuing after the current maximum (reference :154-179).
    The caller's iteration order decides the ids (the reference iterates a `set`: SURVEY.md F5)."""
    vocab = dict(base_vocab)
    nxt = max(vocab.values()) + 1
    for tok in dynamic_tokens:
        if tok not in vocab:
            vocab[tok] = nxt
            nxt += 1
    return vocab


def build_zinc_vocab_on_device(texts, device=None, capacity: int = 1 << 16) -> Dict[str, int]:
    """The vocab pass of trainer/train_ibtt.py:361-372 - fixed table + every token of the train / val / test
    texts that is not in it - with the scan over all texts done by one launch (gtok_vocab_stats_text).  The
    reference gives the unseen tokens ids in the iteration order of a Python `set` of str, which depends on
    PYTHONHASHSEED and cannot be reproduced without fixing it; the DETERMINISTIC stand-in used
    here is first appearance in the corpus (texts in the order given: train, val, test).  Same token set, same id
    range 22.., a different permutation of the dynamic ids - which is why every tokenizer entry point takes the
    vocab as an input: a vocab saved by the reference (its checkpoints embed it) is replayed unchanged."""
    import torch
    from ._root import root as _root
    ops = _root().ops
    vocab, _ = build_fixed_zinc_vocab()
    texts = list(texts)
    if not texts:
        return vocab
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    blob, ptr = ops.pack_texts(texts)
    blob = blob.to(device)
    entries = ops.text_stats_entries(ops.vocab_stats_text(blob, ptr, capacity), blob)
    unseen = sorted((first, tok) for tok, _, first in entries if tok not in vocab)
    return extend_vocab_with_dynamic_tokens(vocab, [tok for _, tok in unseen])


def map_autograph_token_to_fixed_id(autograph_token_id: int, tokenizer_node_idx_offset: int,
                                    tokenizer_edge_idx_offset: int, is_node_type: bool = False,
                                    is_edge_type: bool = False) -> int:
    """SOS/EOS/PAD -> <bos>/<eos>/<pad>; typed tokens through the fixed tables; anything else unchanged
    (reference :186-238; exported there but unused by the trainers)."""
    special = {0: 0, 4: 1, 5: 2}
    if autograph_token_id in special:
        return special[autograph_token_id]
    if is_node_type:
        return get_atom_type_id(autograph_token_id - tokenizer_node_idx_offset)
    if is_edge_type:
        return get_bond_type_id(autograph_token_id - tokenizer_edge_idx_offset + 1)
    return autograph_token_id
