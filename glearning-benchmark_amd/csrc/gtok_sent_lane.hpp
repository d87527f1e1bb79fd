// gtok_sent_lane.hpp — SENT walk, LANE per graph: one wavefront tokenizes 64 small graphs at once.
//
// The wave-per-graph kernels spend ~1 scalar + ~1 vector instruction slot per walk step per graph; a molecule
// has ~25 nodes, so 63 of 64 lanes idle in every vector instruction.  Here every lane runs the whole walk of its
// own graph — the same spec and token stream, bit-exact against oracle/gtok_oracle.c — so one
// vector instruction advances 64 graphs.  The kernel's time follows its instruction count per step and the number
// of waves a SIMD can interleave, so the design goals are: few instructions per step, <= 128 VGPRs (4 waves per
// SIMD, 16 per CU: the 3898 64-graph units of ZINC-full run in ONE round) and <= 10 KB of LDS per wave.
//
// Requires GTOK_CSR_SIMPLE_SYMMETRIC (host-verified: no self loop, no duplicate entry, every (u,v) has its (v,u) —
// any PyG-coalesced undirected graph): the staged CSR rows ARE the adjacency lists and the edge type of a neighbour
// sits next to its id.  LDS per wave — the CSR chunk of the wave's 64 consecutive graphs, packed to bytes:
//   srp   row pointers, scol neighbour ids, seat edge types, snat node types.  A lane indexes its own graph
//   through its offsets; a row's first four neighbours (and their edge types) arrive in ONE ds_read2_b32 each
//   (the aligned dword pair around the row start, shifted into place with v_alignbyte).
//   snat doubles as the node -> visit index table: a node's type is read exactly once, at its first visit, and
//   the same byte then holds its visit index.
// Walk state lives in registers.  `live` — the visited nodes that still own an uncovered edge — is kept through P
// bit-sliced counters (plane p holds bit p of every node's count of unvisited neighbours): a visit decrements the
// counters of all neighbours at once with 2 P 64-bit operations and no memory access.
// Tokens are collected, 16 bits each, in a 64-bit register window; every 4 tokens leave as one 16-byte store to the
// lane's own slab row (lines are written in whole 16-byte pieces, each once); the pad tails of the 64 rows are
// filled cooperatively (coalesced) at the end of the unit.  Limits: maxn <= 64, maxe <= 255 (u8 indices),
// degree < 2^P (the launcher picks P from gtok_csr.max_degree).
// Round 4 (ABI v4): a launch walks (unit, epoch) pairs (gtok_sent_params.epoch_count: K epochs of a split in one launch);
// pairs beyond the first round of resident waves are drawn from a ticket counter in the workgroup's LDS; rows may be
// 16 bits wide (GTOK_SENT_U16: the windows are stored as they stand); the padding of the last units is shared by the
// waves of a CU; slabs beyond the memory-side cache are padded with non-temporal stores.
#pragma once
#include <type_traits>
#include "gtok_sent_reg.hpp"

namespace gtok {

#ifndef GTOK_LANE_SECTOR_GROUPS
#define GTOK_LANE_SECTOR_GROUPS 2
#endif
#ifndef GTOK_LANE_SECTOR_GROUPS_U16
#define GTOK_LANE_SECTOR_GROUPS_U16 2      // 16-bit slab: 2 windows = 8 ids = ONE 16-byte store per burst (4: 32 bytes, two stores)
#endif

struct __attribute__((aligned(4))) U32x2a4 { uint32_t lo, hi; };   // two dwords at a 4-byte aligned LDS address: ds_read2_b32

// k-th (0-based) set bit of w, per lane, k < popcount(w): branch-free halving on popcounts.  (A clear-lowest-bit
// loop runs max-over-lanes(k) times for the whole wave; picks from `live` have k up to the molecule's size.)
// Binary search on prefix popcounts, five instructions per level and no compare: with nk = -k - 1,
// v_bcnt_u32_b32(low `mid` bits of x, nk) = popcount - k - 1 is negative exactly when the k-th bit lies at or above `mid`;
// its sign, spread over the word, ORs the level's bit into the answer.  (The halving version - mask, popcount, compare, two
// selects, subtract per level - was 7 per level and serialised on VCC.)
__device__ __forceinline__ int kth_bit32(uint32_t x, int k) {
  // Priced with profiles/tools/probes/valu_op_cost_probe.hip (SIMD cycles per wave64 instruction, 4 waves per SIMD: add / sub / and /
  // or / xor / lshr / bitop3 ~2.2 - 2.6, everything else - lshl, bfe, bcnt, and_or, perm, ... - 4.2): the low `mid` bits of x
  // are the TOP `mid` bits of its bit reversal, so a right shift (2.2) stands in for the bit-field extract (4.2); the sign
  // of popcount - k - 1 is moved onto the level's bit with a right shift and merged with one bitop3 (was ashr + and_or).
  const uint32_t nk = ~(uint32_t)k;                       // -k - 1
  const uint32_t y = __builtin_bitreverse32(x);
  // One asm statement for the five levels (the compiler closes every asm statement with a wait state of its own), and the
  // first level - base 0: the shift is a constant, the merge a plain and - takes four instructions instead of five.
  uint32_t base, mid, t;
  asm("v_lshrrev_b32 %2, 16, %3\n\t"                    /* the low 16 bits of x */
      "v_bcnt_u32_b32 %2, %2, %4\n\t"                   /* their popcount - k - 1: negative = bit k is above */
      "v_lshrrev_b32 %2, 27, %2\n\t"                    /* sign -> bit 4 */
      "v_and_b32 %0, 16, %2\n\t"
      "v_sub_u32 %1, 24, %0\n\t"                        /* 32 - (base + 8) */
      "v_lshrrev_b32 %2, %1, %3\n\t"                    /* the low base + 8 bits of x */
      "v_bcnt_u32_b32 %2, %2, %4\n\t"
      "v_lshrrev_b32 %2, 28, %2\n\t"
      "v_bitop3_b32 %0, %2, 8, %0 bitop3:0xea\n\t"      /* base | (sign & half) */
      "v_sub_u32 %1, 28, %0\n\t"
      "v_lshrrev_b32 %2, %1, %3\n\t"
      "v_bcnt_u32_b32 %2, %2, %4\n\t"
      "v_lshrrev_b32 %2, 29, %2\n\t"
      "v_bitop3_b32 %0, %2, 4, %0 bitop3:0xea\n\t"
      "v_sub_u32 %1, 30, %0\n\t"
      "v_lshrrev_b32 %2, %1, %3\n\t"
      "v_bcnt_u32_b32 %2, %2, %4\n\t"
      "v_lshrrev_b32 %2, 30, %2\n\t"
      "v_bitop3_b32 %0, %2, 2, %0 bitop3:0xea\n\t"
      "v_sub_u32 %1, 31, %0\n\t"
      "v_lshrrev_b32 %2, %1, %3\n\t"
      "v_bcnt_u32_b32 %2, %2, %4\n\t"
      "v_lshrrev_b32 %2, 31, %2\n\t"
      "v_bitop3_b32 %0, %2, 1, %0 bitop3:0xea"
      : "=&v"(base), "=&v"(mid), "=&v"(t) : "v"(y), "v"(nk));
  return (int)base;
}

__device__ __forceinline__ int kth_bit64(uint64_t w, int k) {
  const uint32_t lo = (uint32_t)w;
  const int c = __popc(lo);
  const bool up = k >= c;
  return kth_bit32(up ? (uint32_t)(w >> 32) : lo, up ? k - c : k) + (up ? 32 : 0);
}

// 16-byte-vector staging of a byte array into LDS: lane_load16 issues the first U vectors of the lane, lane_commit16
// writes them (then whatever lies beyond U * 64 vectors, and the last < 16 bytes) - `src` has the array's element
// alignment only, the LDS destination is 16-byte aligned
template <int U>
__device__ __forceinline__ void lane_load16(const uint8_t *__restrict__ src, int count, int lane, U8x16 (&r)[U]) {
  const int nv = count >> 4;
  const U8x16 *v = reinterpret_cast<const U8x16 *>(src);
#pragma unroll
  for (int j = 0; j < U; ++j) { const int t = lane + j * kWave; r[j] = t < nv ? v[t] : U8x16{0, 0, 0, 0}; }
}
template <int U>
__device__ __forceinline__ void lane_commit16(const uint8_t *__restrict__ src, int count, int lane, const U8x16 (&r)[U], uint8_t *dst) {
  const int nv = count >> 4;
  const U8x16 *v = reinterpret_cast<const U8x16 *>(src);
  U8x16a *d = reinterpret_cast<U8x16a *>(dst);
#pragma unroll
  for (int j = 0; j < U; ++j) { const int t = lane + j * kWave; if (t < nv) d[t] = U8x16a{r[j].a, r[j].b, r[j].c, r[j].d}; }
  _Pragma("clang loop vectorize(disable) unroll(disable)")
  for (int t = lane + U * kWave; t < nv; t += kWave) { const U8x16 x = v[t]; d[t] = U8x16a{x.a, x.b, x.c, x.d}; }
  if (lane < (count & 15)) dst[(nv << 4) + lane] = src[(nv << 4) + lane];
}

// Padding store.  (-DGTOK_SC1_PAD_STORES: `sc1` stores, written through and dropped from the XCD's L2 - tried to keep
// the padding, 56 % of a ZINC slab, from pushing half-written token lines out of L2: the counted write traffic did not
// move (359 vs 353 MB) and the walks got slower behind the write-through stores, 0.120 vs 0.1175 ms.)
// nt: non-temporal stores - for slabs beyond the 256 MB of the memory-side cache the padding (more than half of a molecule
// slab) otherwise pushes the half-written token lines out ahead of their time: 31 k molecules x 21 epochs 0.289 -> 0.190 ms,
// 1 M molecules 0.378 -> 0.345 ms; a slab that fits (ZINC-full: 208 MB) is 1.5 % slower with them, so the launcher decides
// The loops that use it are the tail of every wave's life (the padding of a unit's 64 rows): the padding vector is made
// once per call (pad_vec) and the non-temporal switch is taken outside the store loops - with the vector rebuilt and the
// switch tested at every store the ZINC-full launch was 3 us longer.
typedef int gtok_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gtok_v4i pad_vec(const int32_t *, int pad) { gtok_v4i v = {pad, pad, pad, pad}; asm volatile("" : "+v"(v)); return v; }
// (GTOK_SENT_U16 slab: eight 16-bit ids per 16-byte store)
__device__ __forceinline__ gtok_v4i pad_vec(const uint16_t *, int pad) {
  const int pp = (int)(((uint32_t)pad & 0xFFFFu) * 0x00010001u);
  gtok_v4i v = {pp, pp, pp, pp};
  asm volatile("" : "+v"(v));
  return v;
}
template <bool NT>
__device__ __forceinline__ void store_pad16(void *p, const gtok_v4i &v) {
#if defined(GTOK_SC1_PAD_STORES)
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
#endif
}
// four tokens of a row at once: a 16-byte store into the int32 slab, an 8-byte store into the 16-bit slab
__device__ __forceinline__ void store_tok4(int32_t *p, int t0, int t1, int t2, int t3) { *reinterpret_cast<I32x4 *>(p) = I32x4{t0, t1, t2, t3}; }
__device__ __forceinline__ void store_tok4(uint16_t *p, int t0, int t1, int t2, int t3) {
  *reinterpret_cast<I32x2 *>(p) = I32x2{(int)(((uint32_t)t0 & 0xFFFFu) | ((uint32_t)t1 << 16)), (int)(((uint32_t)t2 & 0xFFFFu) | ((uint32_t)t3 << 16))};
}
// a window of four 16-bit tokens as it stands (the 16-bit slab stores it unchanged: no unpacking)
__device__ __forceinline__ void store_win(int32_t *p, uint64_t w) {
  *reinterpret_cast<I32x4 *>(p) = I32x4{(int)(w & 0xFFFFu), (int)((w >> 16) & 0xFFFFu), (int)((w >> 32) & 0xFFFFu), (int)(w >> 48)};
}
__device__ __forceinline__ void store_win(uint16_t *p, uint64_t w) { *reinterpret_cast<I32x2 *>(p) = I32x2{(int)(uint32_t)w, (int)(uint32_t)(w >> 32)}; }
__device__ __forceinline__ void store_win2(uint16_t *p, uint64_t w0, uint64_t w1) {   // two windows = eight ids = one 16-byte store
  *reinterpret_cast<I32x4 *>(p) = I32x4{(int)(uint32_t)w0, (int)(uint32_t)(w0 >> 32), (int)(uint32_t)w1, (int)(uint32_t)(w1 >> 32)};
}

// inclusive prefix sum over the wave's 64 lanes in the DPP network (no LDS round trips): within rows of 16 lanes, then the
// rows' totals broadcast into the rows behind them
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

constexpr int kLaneWgShared = 256;   // bytes of workgroup-shared LDS behind the waves' slices (per-CU launch)

struct SentLaneArgs {
  gtok_csr g;
  gtok_sent_params p;
  int off_rp, off_col, off_eat, off_nat;   // staged CSR chunk of the wave (u8): row pointers, neighbour ids, edge / node types
  int lds;                                 // bytes of LDS per 64-graph wave
  int maxn;                                // largest graph of the batch (<= 64)
  int cap_r, cap_n, cap_e;                 // staging capacities: row pointers, nodes, entries of one 64-graph chunk
  int32_t *out;
  int ld;
  int32_t *out_len;
  int units;      // units of ONE epoch (64-graph groups, or the unit table of a reordered batch)
  int epochs;     // K >= 1 (gtok_sent_params.epoch_count): the launch walks units x K (unit, epoch) pairs, unit-major
  int unit_mul;   // 0: units in order
  int prio_cut[3];   // reordered batch: units below these ranks (in 64ths of the stored order) run at priority 3 / 2 / 1
  int pad_nt;        // padding leaves with non-temporal stores (slabs larger than the memory-side cache)
  int epoch_major;   // order of the (unit, epoch) pairs: 0 = unit-major (pair v = unit v / K), 1 = epoch-major (pair v = unit v mod units)
  // gtok_sent_packed (NULL otherwise): rows are also appended, 16-byte aligned, to pack_out behind the fill mark pack_state[0]
  void *pack_out;
  int64_t *pack_start;              // [K * G] first id of row (epoch, graph) in pack_out; -1: did not fit
  unsigned long long *pack_state;   // GTOK_PACK_STATE_WORDS words: [0] status bits, [FILL + STRIDE * r] ids used in region r
  int64_t pack_region_cap;          // ids per region (a multiple of 8)
  int pack_regions;                 // a power of two <= GTOK_PACK_REGIONS
  int pack_scratch;                 // GTOK_SENT_PACK_ONLY: `out` is staging space, 64 rows per resident wave
};

// PK: the batch carries the byte-packed rowptr / col mirror (gtok_csr.rowptr8 / col8): a unit is staged with 12
// 16-byte loads per lane, all in flight at once, and no packing instructions
// U16: the GTOK_SENT_U16 slab - rows of 16-bit ids, the token windows stored as they stand
// PACK: gtok_sent_packed - a finished unit's rows are also appended to the packed buffer (an instantiation of its own: the plain
// kernel's register allocation is not to be touched)
template <bool LAB, int P, bool REMAP, bool PK, bool U16, bool PACK = false>
__global__ void __launch_bounds__(1024) sent_lane_kernel(const SentLaneArgs a) {
  using out_t = typename std::conditional<U16, uint16_t, int32_t>::type;
  constexpr int EV = U16 ? 8 : 4;            // ids per 16-byte store
  out_t *const out_base = reinterpret_cast<out_t *>(a.out);
  // one-wave workgroups (batch in dataset order), or ONE 16-wave workgroup per CU (reordered batch: see the unit loop);
  // waves never cooperate either way, each owns a.lds bytes of the workgroup's LDS
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int lane = lane_id(), wave = wave_id();
  unsigned char *smem = smem_all + (size_t)wave * a.lds;
  uint8_t *srp = smem + a.off_rp, *scol = smem + a.off_col, *seat = smem + a.off_eat, *snat = smem + a.off_nat;

  const int lim = a.p.max_len, ld = a.ld, cap = min(lim, ld);
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  const int node_off = idx_off + a.p.max_num_nodes;
  const int edge_off = node_off + a.p.num_node_types;
  // the Philox key lives in VECTOR registers: as scalars the compiler precomputes the ten round keys of both halves
  // (20 SGPRs, spilled, read back with v_readlane in every block); as vectors a round costs two v_add
  uint32_t k0, k1;
  asm volatile("v_mov_b32 %0, %1" : "=v"(k0) : "s"((uint32_t)a.p.seed));
  asm volatile("v_mov_b32 %0, %1" : "=v"(k1) : "s"((uint32_t)(a.p.seed >> 32)));
  const uint32_t epoch0 = (uint32_t)a.p.epoch;
  constexpr bool remap = REMAP;             // folded into the emission constants (host guarantees maxn <= max_num_nodes)
  const int pos_base = remap ? 22 : idx_off;
  const uint64_t T_RESET = remap ? 2 : GTOK_SENT_RESET, T_LADJ = remap ? 2 : GTOK_SENT_LADJ;
  const uint64_t T_RADJ = remap ? 2 : GTOK_SENT_RADJ, T_EOS = remap ? 1 : GTOK_SENT_EOS;
  const int G = a.g.num_graphs, pad = a.p.pad_id;
  const int cap_r = a.cap_r, cap_e = a.cap_e, cap_n = a.cap_n;

  // ---- staging.  A unit's chunk is contiguous in every CSR array; it is loaded with 16-byte vectors (4 ids / 4 row
  // pointers / 16 type bytes per lane and load), packed to bytes and written to LDS in two phases so that at most
  // ~64 staging registers are live: A = row pointers + types, B = neighbour ids.
  constexpr int UR = 8, UC = 16, UE = 4, UN = 2;   // vectors per lane held in registers (chunks beyond: tail loops)
  struct Hdr { int g0, gl, N0, N1; int64_t E0, E1; int nb0, nfull, n, e; int64_t e0; bool valid; int g; };
  auto header = [&](int unit) __attribute__((always_inline)) -> Hdr {
    Hdr h;
    if (a.g.unit_info) {                 // one 32-byte record: the chunk's loads can go out right behind this one round trip
      const int32_t *ui = a.g.unit_info + 8 * (int64_t)unit;
      h.g0 = sload(ui, 0); h.gl = sload(ui, 1); h.N0 = sload(ui, 2); h.N1 = sload(ui, 3);
      h.E0 = (int64_t)(((uint64_t)(uint32_t)sload(ui, 5) << 32) | (uint32_t)sload(ui, 4));
      h.E1 = (int64_t)(((uint64_t)(uint32_t)sload(ui, 7) << 32) | (uint32_t)sload(ui, 6));
    } else {
      if (a.g.unit_ptr) { h.g0 = sload(a.g.unit_ptr, unit); h.gl = sload(a.g.unit_ptr, unit + 1); }   // reordered batch: <= 64 slots
      else { h.g0 = unit * 64; h.gl = min(h.g0 + 64, G); }
      h.N0 = sload(a.g.node_ptr, h.g0); h.N1 = sload(a.g.node_ptr, h.gl);
      h.E0 = sload(a.g.edge_ptr, h.g0); h.E1 = sload(a.g.edge_ptr, h.gl);
    }
    h.valid = h.g0 + lane < h.gl;
    h.nb0 = h.N0; h.nfull = 0; h.e0 = h.E0; h.e = 0;
    h.g = h.g0 + lane;
    if (h.valid) {
      h.nb0 = a.g.node_ptr[h.g0 + lane];
      h.nfull = a.g.node_ptr[h.g0 + lane + 1] - h.nb0;
      h.e0 = a.g.edge_ptr[h.g0 + lane];
      h.e = min((int)(a.g.edge_ptr[h.g0 + lane + 1] - h.e0), a.g.max_edges);
      // the graph's dataset index: output row, out_len / query entry and RNG identity (slot = where its CSR is stored) -
      // asked for here, with the other per-lane loads, not behind the staging fence where its round trip stood alone
      if (a.g.graph_ids) h.g = a.g.graph_ids[h.g0 + lane];
    }
    h.n = min(h.nfull, a.maxn);
    return h;
  };
  auto pack4 = [](const I32x4 &v) -> uint32_t {
    return ((uint32_t)v.x & 255u) | (((uint32_t)v.y & 255u) << 8) | (((uint32_t)v.z & 255u) << 16) | ((uint32_t)v.w << 24);
  };
  struct RegsA { I32x4 rv[UR]; U8x16 ev[UE], nv[UN]; };
  auto load_a = [&](const Hdr &h, RegsA &r) __attribute__((always_inline)) {
    const int nrv = min((h.N1 - h.N0) + (h.gl - h.g0), cap_r) >> 2;
    const I32x4 *rpv = reinterpret_cast<const I32x4 *>(a.g.rowptr + h.N0 + h.g0);
#pragma unroll
    for (int j = 0; j < UR; ++j) { const int t = lane + j * kWave; r.rv[j] = t < nrv ? rpv[t] : I32x4{0, 0, 0, 0}; }
    if (LAB) {
      lane_load16<UE>(a.g.eattr + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.ev);
      lane_load16<UN>(a.g.nattr + h.N0, min(h.N1 - h.N0, cap_n), lane, r.nv);
    }
  };
  auto commit_a = [&](const Hdr &h, const RegsA &r) __attribute__((always_inline)) {
    const int cr = min((h.N1 - h.N0) + (h.gl - h.g0), cap_r);   // graph g's row pointers start at node_ptr[g] + g
    const int32_t *__restrict__ rpc = a.g.rowptr + h.N0 + h.g0;
    const I32x4 *rpv = reinterpret_cast<const I32x4 *>(rpc);
    uint32_t *srp4 = reinterpret_cast<uint32_t *>(srp);
    const int nrv = cr >> 2;
#pragma unroll
    for (int j = 0; j < UR; ++j) { const int t = lane + j * kWave; if (t < nrv) srp4[t] = pack4(r.rv[j]); }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (int t = lane + UR * kWave; t < nrv; t += kWave) srp4[t] = pack4(rpv[t]);
    if (lane < (cr & 3)) srp[(nrv << 2) + lane] = (uint8_t)rpc[(nrv << 2) + lane];
    if (LAB) {
      lane_commit16<UE>(a.g.eattr + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.ev, seat);
      lane_commit16<UN>(a.g.nattr + h.N0, min(h.N1 - h.N0, cap_n), lane, r.nv, snat);
    }
  };
  // packed mirror: everything is bytes
  constexpr int PR = 2, PC = 4;
  struct RegsP { U8x16 rv[PR], cv[PC], ev[UE], nv[UN]; };
  auto load_p = [&](const Hdr &h, RegsP &r) __attribute__((always_inline)) {
    lane_load16<PR>(a.g.rowptr8 + h.N0 + h.g0, min((h.N1 - h.N0) + (h.gl - h.g0), cap_r), lane, r.rv);
    lane_load16<PC>(a.g.col8 + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.cv);
    if (LAB) {
      lane_load16<UE>(a.g.eattr + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.ev);
      lane_load16<UN>(a.g.nattr + h.N0, min(h.N1 - h.N0, cap_n), lane, r.nv);
    }
  };
  auto commit_p = [&](const Hdr &h, const RegsP &r) __attribute__((always_inline)) {
    lane_commit16<PR>(a.g.rowptr8 + h.N0 + h.g0, min((h.N1 - h.N0) + (h.gl - h.g0), cap_r), lane, r.rv, srp);
    lane_commit16<PC>(a.g.col8 + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.cv, scol);
    if (LAB) {
      lane_commit16<UE>(a.g.eattr + h.E0, (int)min(h.E1 - h.E0, (int64_t)cap_e), lane, r.ev, seat);
      lane_commit16<UN>(a.g.nattr + h.N0, min(h.N1 - h.N0, cap_n), lane, r.nv, snat);
    }
  };
  auto stage_b = [&](const Hdr &h) __attribute__((always_inline)) {
    const int ce = (int)min(h.E1 - h.E0, (int64_t)cap_e);
    const int32_t *__restrict__ cc = a.g.col + h.E0;
    const I32x4 *ccv = reinterpret_cast<const I32x4 *>(cc);
    uint32_t *scol4 = reinterpret_cast<uint32_t *>(scol);
    const int ncv = ce >> 2;
    I32x4 cv[UC];
#pragma unroll
    for (int j = 0; j < UC; ++j) { const int t = lane + j * kWave; cv[j] = t < ncv ? ccv[t] : I32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int j = 0; j < UC; ++j) { const int t = lane + j * kWave; if (t < ncv) scol4[t] = pack4(cv[j]); }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (int t = lane + UC * kWave; t < ncv; t += kWave) scol4[t] = pack4(ccv[t]);
    if (lane < (ce & 3)) scol[(ncv << 2) + lane] = (uint8_t)cc[(ncv << 2) + lane];
  };

  // Units are dealt round-robin (a unit's time is the longest of its 64 walks: they are all alike); with 16
  // resident waves per CU a ZINC-full launch gives every wave exactly one unit.
  const int stride = (int)gridDim.x;
  int lw = 0, done_row = -1, done_cnt = 0;  // pad start of this lane's finished row, that row (-1: none), rows of the finished unit
  // pad the tails of a finished unit's 64 rows: four rows per pass, 16 lanes x 16-byte stores on each
  const bool no_pad = (a.p.flags & GTOK_SENT_NO_PAD) != 0, pad_nt = a.pad_nt != 0;
  auto pad_rows_impl = [&](auto nt_tag) __attribute__((always_inline)) {
    constexpr bool NT = decltype(nt_tag)::value;
    const int q = lane & 15;
    const gtok_v4i pv = pad_vec(out_base, pad);
    // (the padding is the tail of a wave's life - of the launch, for the last wave: the two cross-lane reads of the NEXT pass
    // are requested before this pass's stores, so that their LDS round trip does not stand in the chain 16 times over)
    int lr_n = __builtin_amdgcn_ds_bpermute((lane >> 4) << 2, lw);
    int gr_n = __builtin_amdgcn_ds_bpermute((lane >> 4) << 2, done_row);   // (rows of a reordered batch are not neighbours)
    for (int it = 0; it < 16; ++it) {
      const int lr = lr_n, gr = gr_n;
      if (it * 4 >= done_cnt) break;
      const int rn = min(it * 4 + 4, 60) + (lane >> 4);
      lr_n = __builtin_amdgcn_ds_bpermute(rn << 2, lw);
      gr_n = __builtin_amdgcn_ds_bpermute(rn << 2, done_row);
      if (gr >= 0) {
        out_t *__restrict__ rowp = out_base + (int64_t)gr * ld + lr;
        const int nrem = ld - lr, nvec = nrem / EV;
        _Pragma("clang loop vectorize(disable) unroll(disable)")
        for (int t = q; t < nvec; t += 16) store_pad16<NT>(rowp + EV * t, pv);
        if (q < (nrem & (EV - 1))) rowp[nvec * EV + q] = (out_t)pad;
      }
    }
  };
  auto pad_rows = [&]() __attribute__((always_inline)) {
    if (no_pad) return;
    if (pad_nt) pad_rows_impl(std::true_type{});
    else pad_rows_impl(std::false_type{});
  };
#ifdef GTOK_PHASE_TIMING   // profiling build only: cycle stamps per phase, left in the last 8 columns of the unit's first row
  uint64_t ts[5] = {0, 0, 0, 0, 0}, pkt[2] = {0, 0};
  uint32_t rt0 = 0, iters = 0;
  auto stamps_out = [&]() __attribute__((always_inline)) {
    if (!U16 && lane == 0 && ld >= 16 && done_row >= 0) {
      int32_t *row = a.out + (int64_t)done_row * ld + ld - 8;
      row[0] = (int32_t)(ts[1] - ts[0]); row[1] = (int32_t)(ts[2] - ts[1]); row[2] = (int32_t)(ts[3] - ts[2]);
      row[3] = (int32_t)(ts[4] - ts[3]); row[4] = (int32_t)rt0; row[5] = (int32_t)__builtin_amdgcn_s_memrealtime();
      row[6] = (int32_t)iters; row[7] = (int32_t)blockIdx.x;
      if (PACK) { row[6] = (int32_t)pkt[0]; row[7] = (int32_t)pkt[1]; }   // gtok_sent_packed: cycles until the atomic answered / of the copy
    }
  };
#endif
  // A reordered batch stores its units by descending walk length, and a wave is as fast as the company it keeps on its
  // SIMD lets it be (alone ~2.5 k cycles per step, with three others ~5.2 k): every SIMD should hold one unit of each
  // quarter of that order.  HIP gives no say in which one-wave workgroups share a SIMD, but inside ONE workgroup the waves
  // w, w + 4, w + 8, w + 12 share one (observed on gfx950: profiles/tools/probes/wave_simd_probe.hip; speed only) - so the
  // launch is 256 workgroups of 16 waves, one per CU with all of its LDS, and wave w of workgroup b takes, in the FIRST round
  // of 4,096 units, rank (w >> 2) * 1024 + j of the stored order, j = (w & 3) * 256 + b, odd quarters backwards (long with
  // short).  The quarter is also the wave's issue priority in that round: the stragglers-to-be get the slots their SIMD-mates
  // can spare.  (Three waves per SIMD - GTOK_LANE_WG_WAVES=12 - with the rest of the units dealt dynamically were measured: worse.)
  const bool percu = blockDim.x > 64;
  // (GTOK_LANE_WG_WAVES=8: two 8-wave workgroups per CU, each balanced in itself - quarters 0 + 3 or 1 + 2 on every SIMD)
  const int wg_waves = (int)(blockDim.x >> 6), wgs_per_cu = wg_waves == 8 ? 2 : 1;
  const int cus = (int)gridDim.x / wgs_per_cu, wg_type = (int)blockIdx.x % wgs_per_cu, wg_cu = (int)blockIdx.x / wgs_per_cu;
  const int nslots = (int)gridDim.x * wg_waves, qsize = cus * 4;
  const int quarter = wg_waves == 8 ? (wg_type == 0 ? ((wave >> 2) ? 3 : 0) : ((wave >> 2) ? 2 : 1)) : wave >> 2;
  const int jq = (wave & 3) * cus + wg_cu;
  // K epochs in one launch: the launch walks (unit, epoch) pairs - unit-major (the K walks of a unit are neighbours in the
  // deal: it stays sorted by walk length; same CSR chunk: L2 hits), or epoch-major when one epoch fills half of the resident
  // waves or more (the first round is then the tuned one-epoch deal): the launcher picks (a.epoch_major)
  const int K = a.epochs, vunits = a.units * K;
  // Beyond the first round of resident waves (more than nslots pairs: corpora of > 260 k molecules, or K epochs of a
  // smaller one) the pairs are handed out DYNAMICALLY: workgroup b owns the pairs nslots + t * gridDim + b, t = 0, 1, ...
  // (every CU the same mix of lengths, longest first), and a wave that has finished a unit draws the next t from a ticket
  // counter in the workgroup's LDS.  (With the static split of round 3 - every wave one unit of every round, priorities
  // by quarter - the low-priority waves of a SIMD ran their later units alone at half the issue rate: 1 M molecules took
  // 6.5 x the time of 249 k.)
  // the workgroup's shared words behind the waves' slices (kLaneWgShared bytes, zeroed here): [0] the ticket counter,
  // [16 + w] wave w's "last unit published" flag, [32 + w] the next chunk of its padding to hand out (final phase below)
  int *wg_ticket = reinterpret_cast<int *>(smem_all + (size_t)wg_waves * a.lds);
  if (percu) {
    if (threadIdx.x < kLaneWgShared / 4) wg_ticket[threadIdx.x] = 0;
    __syncthreads();
  }
  // first unit: the static deal of the first round (per-CU workgroups), or the one-wave workgroup's own index
  int cursor = percu ? 0 : (a.unit_mul ? (int)blockIdx.x : virtual_block());
  auto spread = [&](int i) -> int { return (a.unit_mul && i < vunits) ? (int)(((uint64_t)(uint32_t)i * (uint32_t)a.unit_mul) % (uint32_t)vunits) : i; };
  int vu = percu ? quarter * qsize + ((quarter & 1) ? qsize - 1 - jq : jq) : spread(cursor);
#ifndef GTOK_LANE_NO_PRIO
  if (percu) {   // the quarter is also the wave's issue priority in the first round
    if (quarter == 0) __builtin_amdgcn_s_setprio(3);
    else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
    else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
#endif
  while (vu < vunits) {
    int unit = vu;
    uint32_t epoch = epoch0;
    if (K > 1) {
      if (a.epoch_major) { const int e_ = vu / a.units; unit = vu - e_ * a.units; epoch += (uint32_t)e_; }
      else { unit = vu / K; epoch += (uint32_t)(vu - unit * K); }
    }
    // ---- stage this unit; the loads of phase A go out ahead of the previous unit's padding stores
#ifdef GTOK_PHASE_TIMING
    const uint64_t ts0_new = __builtin_amdgcn_s_memtime();
    const uint32_t rt0_new = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    const Hdr h = header(unit);
    auto between = [&]() __attribute__((always_inline)) {   // behind the loads, ahead of the LDS writes
      if (done_cnt > 0) pad_rows();
#ifdef GTOK_PHASE_TIMING
      stamps_out();
      ts[0] = ts0_new; rt0 = rt0_new; iters = 0;
#endif
      __builtin_amdgcn_wave_barrier();
    };
    if (PK) {
      RegsP rp;
      load_p(h, rp);
      between();
      commit_p(h, rp);
    } else {
      {
        RegsA ra;
        load_a(h, ra);
        between();
        commit_a(h, ra);
      }
      stage_b(h);
    }
    wave_sync();
#ifdef GTOK_PHASE_TIMING
    ts[1] = __builtin_amdgcn_s_memtime();
#endif
    const bool valid = h.valid;
    const int g = h.g;     // the graph's dataset index (header)
    const int n = h.n, e = h.e;
    const int rbase = (h.nb0 - h.N0) + lane, cbase = (int)(h.e0 - h.E0), nbase = h.nb0 - h.N0;

#ifndef GTOK_LANE_NO_PRIO
    // A unit runs as long as its longest walk, and a launch as long as its slowest unit: the likely stragglers get a
    // higher issue priority than the waves they share a SIMD with, which have slack.  A reordered batch stores its units
    // by descending walk length (the first eighth are the stragglers); otherwise the unit's largest graph predicts it.
    if (percu) {
    } else if (a.g.unit_ptr) {
      const int rank = (int)(((int64_t)unit << 6) / a.units);   // 0..63
      if (rank < a.prio_cut[0]) __builtin_amdgcn_s_setprio(3);
      else if (rank < a.prio_cut[1]) __builtin_amdgcn_s_setprio(2);
      else if (rank < a.prio_cut[2]) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else {
      int mx = n;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
      const int behind = a.maxn - uni(mx);
      if (behind <= 1) __builtin_amdgcn_s_setprio(3);
      else if (behind <= 3) __builtin_amdgcn_s_setprio(2);
      else if (behind <= 5) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    // The unit's body, written once and compiled twice: node sets (visited, rows, counter planes, brackets) as 64-bit
    // words, or as 32-bit words when no graph of the unit has more than 32 nodes - nearly every unit of a molecule corpus,
    // and half the vector instructions of every set operation (a 64-bit AND / OR / shift / popcount is two).
    auto unit_body = [&](auto s32_tag) __attribute__((always_inline)) {
    constexpr bool S32 = decltype(s32_tag)::value;
    using set_t = typename std::conditional<S32, uint32_t, uint64_t>::type;
    constexpr int kSetBits = S32 ? 32 : 64;
    auto bit_of = [](uint32_t i) __attribute__((always_inline)) -> set_t { return (set_t)((set_t)1 << (i & (uint32_t)(kSetBits - 1))); };
    auto popc_of = [](set_t x) __attribute__((always_inline)) -> int { if constexpr (S32) return __popc(x); else return __popcll(x); };
    auto ctz_of = [](set_t x) __attribute__((always_inline)) -> int { if constexpr (S32) return __builtin_ctz(x); else return __builtin_ctzll(x); };
    auto kth_bit_of = [](set_t x, int k) __attribute__((always_inline)) -> int { if constexpr (S32) return kth_bit32(x, k); else return kth_bit64(x, k); };
    // A node's row: bounds, the set of its neighbours, and the first four neighbour ids / edge types as packed
    // bytes (molecules rarely have more; longer rows continue in byte loops).
    struct Row { set_t mask; uint32_t nb4, et4, bm; int rs, deg; };   // bm: byte mask of the valid entries among the first four
    auto load_row = [&](int v) __attribute__((always_inline)) -> Row {
      Row r;
      r.rs = srp[rbase + v];
      r.deg = max(min((int)srp[rbase + v + 1], e) - r.rs, 0);
      const int o = cbase + r.rs;
      const U32x2a4 w = *reinterpret_cast<const U32x2a4 *>(scol + (o & ~3));
      r.nb4 = __builtin_amdgcn_alignbyte(w.hi, w.lo, (uint32_t)(o & 3));
      r.et4 = 0;
      if (LAB) {
        const U32x2a4 t = *reinterpret_cast<const U32x2a4 *>(seat + (o & ~3));
        r.et4 = __builtin_amdgcn_alignbyte(t.hi, t.lo, (uint32_t)(o & 3));
      }
      r.bm = r.deg >= 4 ? 0xFFFFFFFFu : ((1u << (8 * r.deg)) - 1u);
      // entries past the row's end are replaced by its first entry (duplicates are harmless in a set): no per-entry select.
      // v_lshlrev_b64 takes the low 6 bits of its shift operand, so the upper bytes need no masking.
      const uint32_t nv = __builtin_amdgcn_perm(r.nb4, r.nb4, 0x03020100u & r.bm);
      set_t m = bit_of(nv) | bit_of(nv >> 8) | bit_of(nv >> 16) | bit_of(nv >> 24);
      m = r.deg > 0 ? m : (set_t)0;
      for (int k = 4; k < r.deg; ++k) m |= bit_of(scol[o + k]);
      r.mask = m;
      return r;
    };
    // edge type of the listed entry row -> y (y is listed: symmetric adjacency).  Zero-byte search over the four
    // packed ids (the lowest flag of the classic (x - 0x01..) & ~x & 0x80.. test is exact), byte loop beyond.
    auto find_et = [&](const Row &r, uint32_t y) __attribute__((always_inline)) -> uint32_t {
      const uint32_t x = r.nb4 ^ __builtin_amdgcn_perm(y, y, 0u);   // y in all four bytes (a 32-bit multiply is quarter rate)
      const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u & r.bm;
      uint32_t et = 0;
      if (z) {
        et = (r.et4 >> (__builtin_ctz(z) - 7)) & 255u;
      } else {
        const int o = cbase + r.rs;
        for (int k = 4; k < r.deg; ++k) if (scol[o + k] == (uint8_t)y) et = seat[o + k];
      }
      return et;
    };

    // ---- bit-sliced counters: c[p] bit u = bit p of (number of unvisited neighbours of u); starts at the degree
    set_t c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) c[p] = 0;
    if (n > 0) {
      // four nodes per step, from aligned dwords of the lane's row pointers: the byte-wise differences of a
      // non-decreasing byte string need no borrows, so one 32-bit subtraction yields four degrees; bit p of the four
      // is gathered into a nibble of plane p.  (Bits of nodes >= n are garbage and never looked at: `live` is masked
      // by `vis`, decrements only touch neighbours.)
      const uint32_t *w = reinterpret_cast<const uint32_t *>(srp + (rbase & ~3));
      const uint32_t sh = (uint32_t)(rbase & 3);
      uint32_t d0 = w[0], d1 = w[1];
      uint32_t cur4 = __builtin_amdgcn_alignbyte(d1, d0, sh);              // row pointers 0..3
      const int steps = (n + 3) >> 2;
#pragma unroll 4
      for (int i = 0; i < steps; ++i) {
        const uint32_t d2 = w[i + 2];
        const uint32_t nxt4 = __builtin_amdgcn_alignbyte(d2, d1, sh);        // row pointers 4i+4 .. 4i+7
        const uint32_t dg4 = __builtin_amdgcn_alignbyte(nxt4, cur4, 1u) - cur4;   // degrees of nodes 4i .. 4i+3
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const uint32_t x = (dg4 >> p) & 0x01010101u;
          const uint32_t nib = (x | (x >> 7) | (x >> 14) | (x >> 21)) & 15u;
          c[p] |= (set_t)nib << ((4 * i) & (kSetBits - 1));
        }
        d1 = d2; cur4 = nxt4;
      }
    }

#ifdef GTOK_PHASE_TIMING
    ts[2] = __builtin_amdgcn_s_memtime();
#endif
    // ---- walk (per lane; mirrors oracle_sent step for step)
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);
    // (gtok_sent_packed with GTOK_SENT_PACK_ONLY: the slab is only this wave's staging area - 64 rows that every unit of the wave
    // reuses, so that the row stores stay in the caches and what reaches memory is the packed copy)
    const int64_t srow = PACK && a.pack_scratch ? (int64_t)((int)blockIdx.x * (int)(blockDim.x >> 6) + wave) * 64 + lane
                                                 : (int64_t)(epoch - epoch0) * G + g;
    out_t *__restrict__ orow = out_base + srow * ld;
    set_t vis = 0, live = 0;
    uint64_t wlo = 0;
    int nvis = 0, pos = 0, fl = 0, d = 0, cur = 0;

    // token window: tokens fl .. pos-1 of the row sit in wlo, 16 bits each (pos - fl <= 3 between appends).  A full
    // window (4 tokens) joins the groups of its 16-token SECTOR (pg0..pg2: 64 bytes of the row); the sector leaves with
    // four back-to-back 16-byte stores, so every 64-byte piece of a row reaches the L2 complete and within one burst.
    // (One 16-byte store per window left each row line half written for ~15 us at a time, and the open lines of all
    // resident lanes - 32 MB - are the whole L2: 2.8 bytes were counted at the L2's memory side per token byte.)
    constexpr int SG = U16 ? GTOK_LANE_SECTOR_GROUPS_U16 : GTOK_LANE_SECTOR_GROUPS;   // windows per store burst: 4 = 16 tokens, 2 = 8, 1 = every window on its own
    uint64_t pg[SG > 1 ? SG - 1 : 1];
#pragma unroll
    for (int j = 0; j < SG - 1; ++j) pg[j] = 0;
    auto tok_of = [](uint64_t w, int i) __attribute__((always_inline)) -> int { return (int)((w >> (i << 4)) & 0xFFFFu); };
    auto put4 = [&](int at, uint64_t w) __attribute__((always_inline)) {
#ifndef GTOK_ABLATE_STORES   // (profiling builds, profiles/tools/lane_ablate.sh: a phase is cut out - wrong tokens - and the time it took shows)
      store_win(orow + at, w);
#else
      if (w == 0x123456789ABCDEFull) orow[at] = 1;
#endif
    };
    auto group_of = [&](int j, uint64_t w) __attribute__((always_inline)) -> uint64_t {   // group j of the open burst (j = SG-1: w)
      uint64_t r = w;
#pragma unroll
      for (int k = 0; k < SG - 1; ++k) r = j == k ? pg[k] : r;
      return r;
    };
    auto flush = [&](uint64_t w) __attribute__((always_inline)) {
      const int gi = (fl >> 2) & (SG - 1);
#pragma unroll
      for (int k = 0; k < SG - 1; ++k) pg[k] = gi == k ? w : pg[k];
      if (gi == SG - 1) {
        const int sb = fl - 4 * (SG - 1);
        if (fl + 4 <= cap) {
          if constexpr (U16 && (SG & 1) == 0) {      // 16-bit slab: two windows per 16-byte store, nothing to unpack
#ifndef GTOK_ABLATE_STORES
            auto win = [&](int j) __attribute__((always_inline)) -> uint64_t { return j < SG - 1 ? pg[j < SG - 1 ? j : 0] : w; };
#pragma unroll
            for (int k = 0; k < SG; k += 2) store_win2(orow + sb + 4 * k, win(k), win(k + 1));
#else
            if (w == 0x123456789ABCDEFull) orow[sb] = 1;
#endif
          } else {
#pragma unroll
            for (int k = 0; k < SG - 1; ++k) put4(sb + 4 * k, pg[k]);
            put4(fl, w);
          }
        } else {                                 // the row's cut (max_len or a narrow slab) falls inside this burst
          for (int j = 0; j < 4 * SG && sb + j < cap; ++j) orow[sb + j] = (out_t)tok_of(group_of(j >> 2, w), j & 3);
        }
      }
    };
    auto append = [&](uint64_t val, int cnt) __attribute__((always_inline)) {   // cnt <= 4 tokens, lowest first
      const int s = (pos - fl) << 4;
      wlo |= val << s;
      const uint64_t over = (val >> (63 - s)) >> 1;
      pos += cnt;
      if (pos - fl >= 4) { flush(wlo); wlo = over; fl += 4; }
    };
    // decision d uses word d&3 of Philox block d>>2; the block is kept as two packed 64-bit values and the word
    // is extracted with mask arithmetic (a select chain over the captured words makes the compiler select
    // ADDRESSES and park them in scratch: two memory round trips per draw)
    // decision d uses word d&3 of Philox block d>>2.  All lanes of a unit start together and draw once per step, so
    // d is the same in every active lane: the block is refreshed by the whole wave every fourth step and the word is
    // picked with wave-uniform selects (per-lane selection cost a 64-bit mask-and-shift sequence per draw).
    uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
    auto below = [&](uint32_t nchoices) __attribute__((always_inline)) -> uint32_t {
      const int w = uni(d) & 3;
      if (w == 0) {
        uint32_t o[4];
#ifndef GTOK_ABLATE_PHILOX
        philox4x32_10((uint32_t)(d >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
#else
        o[0] = (uint32_t)d * 2654435761u + gid_lo; o[1] = o[0] ^ k0; o[2] = o[0] + k1; o[3] = o[1] + epoch;
#endif
        pw0 = o[0]; pw1 = o[1]; pw2 = o[2]; pw3 = o[3];
      }
      const uint32_t x = w == 0 ? pw0 : (w == 1 ? pw1 : (w == 2 ? pw2 : pw3));
      ++d;
      return __umulhi(x, nchoices);
    };

    if (valid) {
      append((uint64_t)GTOK_SENT_SOS, 1);
      if (n > 0) {
        Row rc{0, 0, 0, 0, 0};   // row of cur, carried from step to step (empty before the first visit)
        const set_t nodes = n >= kSetBits ? ~(set_t)0 : (set_t)(((set_t)1 << (n & (kSetBits - 1))) - 1);
        // Every step draws exactly one decision and touches exactly one node, so the draw, the pick, the row load
        // and the token group are written ONCE and the step's kind only selects operands.  (With one copy per
        // kind, a wave whose lanes are in different kinds - nearly every step - runs every copy.)
        while (pos < lim) {
#ifdef GTOK_PHASE_TIMING
          ++iters;
#endif
          const set_t row = rc.mask & ~vis;
          // 0: extend the trail over an uncovered edge (always towards an unvisited node); 1: dead end, restart from
          // a visited node that still owns uncovered edges; 2: another component or an isolated node
          const int kind = row ? 0 : (live ? 1 : 2);
          if (kind == 2 && nvis >= n) break;
          const set_t set = kind == 0 ? row : (kind == 1 ? live : (set_t)(~vis & nodes));
#ifndef GTOK_ABLATE_PICK
          const int pick = kth_bit_of(set, (int)below((uint32_t)popc_of(set)));
#else
          const int pick = ctz_of(set) + (int)(below((uint32_t)popc_of(set)) >> 31);
#endif
          // the node's byte is asked for first: its round trip then runs beside the edge-type search and the row load
          const uint32_t xb = snat[nbase + pick];              // node type (first visit) or visit index (kind 1)
          uint32_t et = 0;
          if (LAB && kind == 0) et = find_et(rc, (uint32_t)pick);   // type of the listed entry cur -> pick
          rc = load_row(pick);                                 // in place: the old row is not needed past find_et above
          const bool first = kind != 1;
          const uint32_t my = (uint32_t)nvis;
          snat[nbase + pick] = (uint8_t)(first ? my : xb);     // from now on this byte is the node's visit index (kind 1: it already is)
          // ---- the step's token group: [edge type | RESET] position [node type], 16 bits per token
          {
            const uint32_t tpos = (uint32_t)pos_base + (first ? my : xb);
            uint32_t ta = (uint32_t)T_RESET;
            bool has_a = kind == 1 || (kind == 2 && nvis > 0);   // (the walk's first node is a component start without RESET)
            uint32_t lo = tpos, hi = 0;                          // token pair, then the third token
            int cnt = 1;
            if (LAB) {
              if (kind == 0) { ta = remap ? remap_edge_type_u(et, (uint32_t)edge_off) : (uint32_t)edge_off + et; has_a = true; }
              const uint32_t ty = remap ? remap_node_type_u(xb, (uint32_t)node_off, (uint32_t)a.p.num_node_types) : (uint32_t)node_off + xb;
              lo |= first ? ty << 16 : 0u;
              cnt += first;
            }
            if (has_a) { hi = lo >> 16; lo = ta | (lo << 16); ++cnt; }
            append(((uint64_t)hi << 32) | lo, cnt);
          }
          // ---- first visit: neighbours lose an unvisited neighbour; already visited neighbours other than the
          // trail's predecessor are this node's bracket
          const set_t S = first ? rc.mask : (set_t)0;
          set_t M = S & vis & ~(kind == 0 ? bit_of((uint32_t)cur) : (set_t)0);
          {
            set_t b = S, nz = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) { const set_t t = c[p]; c[p] = t ^ b; b &= ~t; nz |= c[p]; }
            vis |= bit_of((uint32_t)pick);
            live = vis & nz;
          }
          nvis += first;
#ifdef GTOK_ABLATE_BRACKET
          M = 0;
#endif
          if (M) {   // LADJ, members by ascending visit index ([edge type] position), RADJ
            bool head = true;
            do {
              set_t t = M;
              int bu = 0, bv = 256;
              do {
                const int u = ctz_of(t);
                t &= t - 1;
                const int vx = snat[nbase + u];
                if (vx < bv) { bv = vx; bu = u; }
              } while (t);
              M &= ~bit_of((uint32_t)bu);
              uint32_t lo = (uint32_t)(pos_base + bv), hi = 0;
              int cnt = 1;
              if (LAB) {
                const uint32_t at = find_et(rc, (uint32_t)bu);
                lo = (remap ? remap_edge_type_u(at, (uint32_t)edge_off) : (uint32_t)edge_off + at) | (lo << 16);
                cnt = 2;
              }
              if (head) { hi = lo >> 16; lo = (uint32_t)T_LADJ | (lo << 16); ++cnt; head = false; }
              uint64_t val = ((uint64_t)hi << 32) | lo;
              if (!M) { val |= T_RADJ << (cnt << 4); ++cnt; }
              append(val, cnt);
            } while (M);
          }
          cur = pick;
        }
      }
      append(T_EOS, 1);
    }
#ifdef GTOK_PHASE_TIMING
    ts[3] = __builtin_amdgcn_s_memtime();
#endif
    // ---- end of the row: what is still in the window, the query tail (trainer/train_agtt.py:257-267: after the
    // trail, original node ids, not remapped), and pad up to the next multiple of 4
    const int len = min(pos, lim);
    int tot = len;
    int q0 = 0, q1 = 0, q2 = 0;
    if (valid && a.p.query) {
      q0 = idx_off + h.nfull; q1 = idx_off + a.p.query[2 * (int64_t)g]; q2 = idx_off + a.p.query[2 * (int64_t)g + 1];
      tot = len + 3;
    }
    int padfrom = 0;                             // where the cooperative padding of this lane's row starts
    const int orow_idx = (int)(epoch - epoch0) * G + g;   // row of the [K, G, ld] slab (K * G < 2^30: checked by the launcher)
    if (valid) {
      a.out_len[orow_idx] = tot;
      const int sb = fl & ~(4 * SG - 1);         // [sb, fl): groups of the open burst, [fl, pos): the window
      if (!a.p.query && pos <= cap && (ld & 3) == 0) {
        // common case (row not cut, no query tail): the open burst, the window and the padding up to the next 16-token
        // boundary leave as one run of 16-byte stores - the row's last 64-byte sector is written whole as well
        const int s16 = fl & ~15;
        padfrom = min(ld, (len + 15) & ~15);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int at = s16 + 4 * k;
          if (at >= sb && at < padfrom) {
            const uint64_t w = at == fl ? wlo : (at > fl ? 0ull : group_of((at - sb) >> 2, 0));
            store_tok4(orow + at, at + 0 < len ? tok_of(w, 0) : pad, at + 1 < len ? tok_of(w, 1) : pad,
                       at + 2 < len ? tok_of(w, 2) : pad, at + 3 < len ? tok_of(w, 3) : pad);
          }
        }
      } else {
        padfrom = min(ld, (tot + EV - 1) & ~(EV - 1));
        for (int i = min(sb, len); i < padfrom; ++i) {
          int v = pad;
          if (i < len) {
            if (i < sb) continue;                // already written by a burst
            v = tok_of(i >= fl ? wlo : group_of((i - sb) >> 2, 0), i & 3);
          } else if (i < tot) {
            v = i == len ? q0 : (i == len + 1 ? q1 : q2);
          }
          orow[i] = (out_t)v;
        }
      }
    }

    lw = padfrom;
    done_row = valid ? (int)srow : -1;
    done_cnt = h.gl - h.g0;
    // ---- gtok_sent_packed: the unit's rows are appended to the packed buffer as well, each from a 16-byte boundary.  The
    // buffer is cut into a.pack_regions equal regions with a fill mark each (one mark for everybody serves ~50-90 units per
    // microsecond: measured, it cost as much as the walk); pair vu appends to region vu mod regions - the units are stored by
    // walk length, so every region receives the same mix and fills evenly - and learns where from ONE atomic add, issued behind
    // the copy's first loads.  The copy is cooperative: 16 lanes per row, four rows per pass (whole lines of the rows this wave
    // has just written).  Rows are ordered by completion: row_start says where each went.
    // (Nothing of this may stay live across the walk: the register file is full.)
    if constexpr (PACK) {
#ifdef GTOK_PHASE_TIMING
      const uint64_t tp0 = __builtin_amdgcn_s_memtime();
      uint64_t tp1 = tp0;
#endif
      const int nid = valid ? min(tot, ld) : 0;
      const int np = (nid + EV - 1) / EV;
      const int inc = wave_incl_scan_dpp(np);
      const int total = __builtin_amdgcn_readlane(inc, kWave - 1);
      const int region = vu & (a.pack_regions - 1);
      const int cnt = h.gl - h.g0;
      const int q = lane & 15, grp = lane >> 4;
      const int my_row = valid ? (int)srow : -1, my_offnp = ((inc - np) << 8) | np;   // (np <= 255: the launcher checks ld)
      const int my_row0 = __builtin_amdgcn_readfirstlane((int)srow);               // (a row of this unit: lane 0 is always valid)
      // The rows read below were written by this wave (other lanes of it).  A CU's stores and loads go through the one vector cache,
      // which must keep a work-item's own store -> load coherent (a write hit updates or drops the line) and does so per address,
      // not per lane; behind it both travel the same in-order path to the L2.  So plain loads see the rows - also where
      // GTOK_SENT_PACK_ONLY reads the same staging rows unit after unit.  (Measured alternatives that do not rely on this: an
      // agent-scope acquire fence or a bare buffer_inv before the loads +2-13 us per epoch, sc1 loads +11 us per epoch at K = 16.)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      // A sweep = 16-byte piece 16 sweep + q of every row, in two batches of eight passes of four rows (what the register file has
      // room for at a unit's end; variants measured slower or equal: the copy behind the next unit's staging loads - its state
      // lives across the walk and spills -, double-buffered batches of four, passes 8-15 landed in LDS by global_load_lds)
      constexpr int KF = 8;
      bool fits = true, placed = false;
      gtok_v4i *dstbase = nullptr;
      for (int sweep = 0;; ++sweep) {
        const int t = sweep * 16 + q;
        _Pragma("clang loop unroll(disable)")
        for (int b0 = 0; b0 < 16; b0 += KF) {
          if (b0 * 4 >= cnt) break;
          gtok_v4i v[KF];
          int doff[KF];
#pragma unroll
          for (int k = 0; k < KF; ++k) {
            const int rn = ((b0 + k) * 4 + grp) << 2;
            const int r_idx = __builtin_amdgcn_ds_bpermute(rn, my_row);
            const int r_on = __builtin_amdgcn_ds_bpermute(rn, my_offnp);
            const bool on = r_idx >= 0 && t < (r_on & 255);
            doff[k] = on ? (r_on >> 8) + t : -1;
            // (an address is always formed - a row of this unit for the lanes that have nothing to fetch - so that the loads leave
            // back to back instead of each behind a branch of its own)
#ifndef GTOK_PACK_ABLATE_COPY
            v[k] = reinterpret_cast<const gtok_v4i *>(out_base + (int64_t)(on ? r_idx : my_row0) * ld)[on ? t : 0];
#else
            v[k] = gtok_v4i{r_idx, r_on, t, k};
#endif
          }
          if (!placed) {
            // the fill mark is asked BEHIND the first loads: one wait covers the unit's row stores draining, the loads and the atomic
            placed = true;
            unsigned long long base = 0;
#ifndef GTOK_PACK_ABLATE_ATOMIC
            if (lane == 0)
              base = __hip_atomic_fetch_add(a.pack_state + GTOK_PACK_STATE_FILL + GTOK_PACK_STATE_STRIDE * region, (unsigned long long)total * EV,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
#else
            base = (unsigned long long)(vu / a.pack_regions) * 64 * 96;
#endif
            fits = base + (unsigned long long)total * EV <= (unsigned long long)a.pack_region_cap;
            const int64_t rstart = (int64_t)region * a.pack_region_cap + (int64_t)base;
            dstbase = reinterpret_cast<gtok_v4i *>(reinterpret_cast<out_t *>(a.pack_out) + rstart);
            if (valid) a.pack_start[orow_idx] = fits ? rstart + (int64_t)(inc - np) * EV : (int64_t)-1;
            if (!fits && lane == 0 && total > 0) atomicOr(a.pack_state, 2ull);
#ifdef GTOK_PHASE_TIMING
            tp1 = __builtin_amdgcn_s_memtime();
#endif
          }
#ifdef GTOK_PACK_ABLATE_COPY
          if (total == 0x7654321)
#endif
          if (fits) {
#pragma unroll
            for (int k = 0; k < KF; ++k) if (doff[k] >= 0) dstbase[doff[k]] = v[k];
          }
        }
        if (__ballot(np > 16 * (sweep + 1)) == 0) break;
      }
#ifdef GTOK_PHASE_TIMING
      pkt[0] = tp1 - tp0; pkt[1] = __builtin_amdgcn_s_memtime() - tp1;
#endif
    }
#ifdef GTOK_PHASE_TIMING
    ts[4] = __builtin_amdgcn_s_memtime();
#endif
    };
    if (__ballot(valid && n > 32) == 0) unit_body(std::true_type{});
    else unit_body(std::false_type{});
    // ---- the next unit
    if (percu) {
      int t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(wg_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      vu = nslots + uni(t) * (int)gridDim.x + (int)blockIdx.x;
#ifndef GTOK_LANE_NO_PRIO
      __builtin_amdgcn_s_setprio(0);   // dynamic rounds: whoever is free takes the longest unit left
#endif
    } else {
      cursor += stride;
      vu = spread(cursor);
    }
  }
  // ---- the last unit's padding.  It is the tail of the wave's life - for the last wave of the launch, of the launch: 16
  // passes of cross-lane reads and stores, ~3 us in which the rest of the CU idles.  In the per-CU launch the waves of a
  // workgroup therefore pad TOGETHER: each publishes its last unit's (row, pad start) pairs in its own LDS slice (free now)
  // and then takes 4-row chunks - of whichever wave has published - until all 16 have published and every chunk is taken.
  // Waves that finish early wait (s_sleep) for the late ones, whose padding is then shared sixteen ways.
  if (percu && !no_pad) {
    int *pub = reinterpret_cast<int *>(smem);
    int *flags = wg_ticket + 16, *next = wg_ticket + 32;
    wave_sync();                                           // every lane is through with the staged chunk
    pub[lane] = done_cnt > 0 ? done_row : -1;
    pub[kWave + lane] = lw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(flags + wave, 1 + done_cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const gtok_v4i pv = pad_vec(out_base, pad);
    const int q = lane & 15;
    for (;;) {
      int fl = 0, nx = 0;
      if (lane < wg_waves) {
        fl = __hip_atomic_load(flags + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        nx = __hip_atomic_load(next + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const int nch = (fl + 2) >> 2;                       // chunks of 4 rows in fl - 1 rows (none while fl == 0)
      const uint64_t avail = __ballot(lane < wg_waves && nx < nch);
      const uint64_t pending = __ballot(lane < wg_waves && fl == 0);
      if (avail) {
        const int w = __builtin_ctzll(avail);
        int c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(next + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = uni(c);
        if (c < __builtin_amdgcn_readlane(nch, w)) {
          const int *pw = reinterpret_cast<const int *>(smem_all + (size_t)w * a.lds);
          const int r = c * 4 + (lane >> 4);
          const int gr = pw[r], lr = pw[kWave + r];
          if (gr >= 0) {
            out_t *__restrict__ rowp = out_base + (int64_t)gr * ld + lr;
            const int nrem = ld - lr, nvec = nrem / EV;
            if (pad_nt) {
              _Pragma("clang loop vectorize(disable) unroll(disable)")
              for (int t = q; t < nvec; t += 16) store_pad16<true>(rowp + EV * t, pv);
            } else {
              _Pragma("clang loop vectorize(disable) unroll(disable)")
              for (int t = q; t < nvec; t += 16) store_pad16<false>(rowp + EV * t, pv);
            }
            if (q < (nrem & (EV - 1))) rowp[nvec * EV + q] = (out_t)pad;
          }
        }
        continue;
      }
      if (!pending) break;
#ifndef GTOK_COOP_SLEEP
#define GTOK_COOP_SLEEP 16
#endif
      __builtin_amdgcn_s_sleep(GTOK_COOP_SLEEP);   // ~1 k cycles between looks: a waiting wave must not take issue slots from the walks still running
    }
  } else if (done_cnt > 0) {
    pad_rows();
  }
#ifdef GTOK_PHASE_TIMING
  stamps_out();
#endif
}

}  // namespace gtok
