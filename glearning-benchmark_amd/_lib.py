"""ctypes binding of csrc/libgtok.so (the C ABI declared in include/gtok.h).

The product path has no CPU fallback: if the HIP library is missing or no
gfx950 device is visible, every op raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# GTOK_LIB: load another build of the same ABI (profiling builds of profiles/tools, e.g. -DGTOK_PHASE_TIMING)
LIB_PATH = os.environ.get("GTOK_LIB") or os.path.join(_HERE, "csrc", "libgtok.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("gtok_sent.hip", "gtok_ibtt.hip", "gtok_rows.hip", "gtok_csr.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "gtok_common.hpp"), os.path.join(_HERE, "csrc", "gtok_sent_blane.hpp"), os.path.join(_HERE, "csrc", "gtok_sent_reg.hpp"), os.path.join(_HERE, "csrc", "gtok_sent_lds.hpp"),
           os.path.join(_HERE, "csrc", "gtok_sent_lane.hpp"),
           os.path.join(_ROOT, "include", "gtok.h")]

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


class GtokCsr(ctypes.Structure):
    _fields_ = [
        ("num_graphs", ctypes.c_int32), ("max_nodes", ctypes.c_int32),
        ("max_edges", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("node_ptr", ctypes.c_void_p), ("edge_ptr", ctypes.c_void_p),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p),
        ("eorder", ctypes.c_void_p), ("nattr", ctypes.c_void_p),
        ("eattr", ctypes.c_void_p),
        ("chunk_nodes", ctypes.c_int32), ("chunk_edges", ctypes.c_int32),
        ("max_degree", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("rowptr8", ctypes.c_void_p), ("col8", ctypes.c_void_p),
        ("adj_rows", ctypes.c_void_p), ("adj_planes", ctypes.c_void_p), ("lane_order", ctypes.c_void_p),
        ("adj_words", ctypes.c_int32), ("adj_max_degree", ctypes.c_int32),
        ("graph_ids", ctypes.c_void_p), ("unit_ptr", ctypes.c_void_p), ("num_units", ctypes.c_int32), ("reserved2", ctypes.c_int32),
        ("unit_info", ctypes.c_void_p),
    ]


CSR_SIMPLE_SYMMETRIC = 1
SENT_NO_PAD = 1
SENT_U16 = 2
SENT_PACK_ONLY = 4


class GtokCsrSorted(ctypes.Structure):
    """include/gtok.h: gtok_csr_sorted - the output arrays of gtok_csr_lane_sort."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("graph_ids", "node_ptr", "edge_ptr", "rowptr", "col", "nattr", "eattr", "rowptr8", "col8",
                                               "unit_ptr", "unit_info", "info")]


class GtokVocabTable(ctypes.Structure):
    _fields_ = [
        ("capacity", ctypes.c_int32), ("pad_id", ctypes.c_int32),
        ("key_off", ctypes.c_void_p), ("key_len", ctypes.c_void_p),
        ("id", ctypes.c_void_p), ("key_bytes", ctypes.c_void_p),
    ]


class GtokSentParams(ctypes.Structure):
    _fields_ = [
        ("max_num_nodes", ctypes.c_int32), ("labeled", ctypes.c_int32),
        ("num_node_types", ctypes.c_int32), ("num_edge_types", ctypes.c_int32),
        ("max_len", ctypes.c_int32), ("remap_zinc", ctypes.c_int32),
        ("pad_id", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("seed", ctypes.c_uint64), ("epoch", ctypes.c_uint64),
        ("graph_base", ctypes.c_int64), ("query", ctypes.c_void_p),
        ("epoch_count", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


# every symbol include/gtok.h declares: (restype, argtypes)
_I, _P = ctypes.c_int32, ctypes.c_void_p
SYMBOLS = {
    "gtok_ibtt_zinc": (_I, [ctypes.POINTER(GtokCsr), _P, _I, _I, _I, _P, _I, _P, _P]),
    "gtok_ibtt_synth": (_I, [ctypes.POINTER(GtokCsr), _P, _I, _P, _I, _I, _P, _I, _P, _P]),
    "gtok_text_to_ids": (_I, [_P, _P, _I, ctypes.POINTER(GtokVocabTable), _I, _I, _P, _I, _P, _P]),
    "gtok_sent": (_I, [ctypes.POINTER(GtokCsr), ctypes.POINTER(GtokSentParams), _P, _I, _P, _P]),
    "gtok_sent_packed": (_I, [ctypes.POINTER(GtokCsr), ctypes.POINTER(GtokSentParams), _P, _I, _P, _P, ctypes.c_int64, _P, _P, _P]),
    "gtok_sent_pack_scratch_rows": (ctypes.c_int64, [_P]),
    "gtok_unpack_rows_at": (_I, [_P, _I, _P, _P, ctypes.c_int64, _I, ctypes.c_int64, ctypes.c_int64, _I, _P, _I, _I, _P, _P]),
    "gtok_sent_decode": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "gtok_remap_zinc": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "gtok_collate": (_I, [_P, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    "gtok_parse_graph_text": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gtok_count_edge_tokens": (_I, [_P, _P, _I, _P, _P]),
    "gtok_find_token": (_I, [_P, _I, _I, ctypes.c_int64, _P, _P]),
    "gtok_vocab_stats_synth": (_I, [ctypes.POINTER(GtokCsr), _P, ctypes.c_int64, _I, _P, _P, _P]),
    "gtok_vocab_stats_text": (_I, [_P, _P, _I, ctypes.c_int64, _I, _P, _P, _P, _P, _P, _P]),
    "gtok_csr_check": (_I, [ctypes.POINTER(GtokCsr), _P, _P]),
    "gtok_csr_lane_sort_workspace": (ctypes.c_int64, [_I]),
    "gtok_csr_lane_sort": (_I, [ctypes.POINTER(GtokCsr), _I, _I, ctypes.POINTER(GtokCsrSorted), _P, ctypes.c_int64, _P]),
    "gtok_csr_adjbits": (_I, [ctypes.POINTER(GtokCsr), _I, _P, _P, _P, _P]),
    "gtok_csr_pack8": (_I, [ctypes.POINTER(GtokCsr), ctypes.c_int64, ctypes.c_int64, _P, _P, _P]),
    "gtok_row_offsets": (_I, [_P, ctypes.c_int64, _I, _I, _P, _P]),
    "gtok_pack_rows": (_I, [_P, _I, _P, ctypes.c_int64, _P, _I, _P, ctypes.c_int64, _P, _P]),
    "gtok_unpack_rows": (_I, [_P, _I, _P, _P, ctypes.c_int64, _I, ctypes.c_int64, _I, _P, _I, _P]),
    "gtok_collate_packed": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P]),
    "gtok_unpack_rows_checked": (_I, [_P, _I, _P, _P, ctypes.c_int64, _I, ctypes.c_int64, ctypes.c_int64, _I, _P, _I, _P, _P]),
    "gtok_unpack_rows_u16": (_I, [_P, _I, _P, _P, ctypes.c_int64, _I, ctypes.c_int64, ctypes.c_int64, _I, _P, _I, _P, _P]),
    "gtok_pack_rows_u16": (_I, [_P, _I, _P, ctypes.c_int64, _P, _I, _P, ctypes.c_int64, _P, _P]),
    "gtok_pack_rows_scan": (_I, [_P, _I, _I, _P, ctypes.c_int64, _I, _I, _P, ctypes.c_int64, _P, _P, _P]),
    "gtok_collate_batch": (_I, [_P, _I, _P, _P, _I, _P, _I, ctypes.c_int64, _I, _P, _P, _I, _P, _I, _P, _P]),
    "gtok_collate_epoch_plan": (_I, [_P, _I, _P, ctypes.c_int64, _I, _P, _P, _P]),
    "gtok_collate_epoch": (_I, [_P, _I, _P, _P, _I, _P, ctypes.c_int64, _I, _I, _P, _P, _P, _P, ctypes.c_int64, _P]),
    "gtok_ids_to_text": (_I, [_P, _I, _P, ctypes.c_int64, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gtok_zinc_text_tails": (_I, [_P, _P, ctypes.c_int64, _I, _P, _P, _P, _P, _P]),
    "gtok_sent_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(GtokCsr), ctypes.POINTER(GtokSentParams)]),
    "gtok_ibtt_zinc_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(GtokCsr)]),
    "gtok_version": (_I, []),
    "gtok_target": (ctypes.c_char_p, []),
}

ERRORS = {-1: "GTOK_E_INVAL", -2: "GTOK_E_TOO_LARGE", -3: "GTOK_E_LAUNCH", -4: "GTOK_E_NO_DEVICE", -5: "GTOK_E_GRAPH_SLOTS", -6: "GTOK_E_UNSUPPORTED"}
E_UNSUPPORTED = -6


class GtokError(RuntimeError):
    pass


def _includes(src: str):
    """Project headers a source includes (directly), for the rebuild check."""
    out = []
    with open(src) as f:
        for line in f:
            if line.startswith('#include "'):
                name = line.split('"')[1]
                for d in (os.path.join(_HERE, "csrc"), os.path.join(_ROOT, "include")):
                    if os.path.exists(os.path.join(d, name)):
                        out.append(os.path.join(d, name))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/libgtok.so (hipcc cross-compiles without a GPU): one object per
    source under csrc/_obj/ (compiled side by side, only the stale ones), then one link."""
    deps = SOURCES + HEADERS
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    common = os.path.join(_HERE, "csrc", "gtok_common.hpp")
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        srcdeps = [src] + _includes(src) + _includes(common)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in srcdeps):
            continue
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c",
               "-I" + os.path.join(_ROOT, "include"), "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None
_lib_version = 0


ABI_VERSION = 6     # include/gtok.h: GTOK_ABI_VERSION


def lib() -> ctypes.CDLL:
    """Load libgtok.so; raise loudly when it has not been built."""
    global _lib, _lib_version
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GtokError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        # GTOK_LIB (A/B timing against the build of an earlier git ref, profiles/tools/ab_build.sh): an ABI v3 / v4 library reads a
        # prefix of today's structs and lacks the newer entry points - accepted for the calls both have, nothing else
        # (library_version() gates the inputs an older library would silently ignore: ops.sent)
        older_ok = bool(os.environ.get("GTOK_LIB")) and 3 <= h.gtok_version() < ABI_VERSION
        for name, (res, args) in SYMBOLS.items():
            if older_ok and not hasattr(h, name):
                continue
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.gtok_version() != ABI_VERSION and not older_ok:      # struct layouts below would not match the library's
            raise GtokError(f"{LIB_PATH} has ABI version {h.gtok_version()}, this binding needs {ABI_VERSION}: rebuild it "
                            "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib, _lib_version = h, int(h.gtok_version())
    return _lib


def library_version() -> int:
    """gtok_version() of the loaded library (== ABI_VERSION unless GTOK_LIB points at an older build)."""
    lib()
    return _lib_version


def check(code: int, what: str) -> None:
    if code != 0:
        raise GtokError(f"{what} failed: {ERRORS.get(code, code)}")
