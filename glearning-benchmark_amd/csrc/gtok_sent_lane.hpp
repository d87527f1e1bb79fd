// gtok_sent_lane.hpp — SENT walk, LANE per graph: one wavefront tokenizes 64 small graphs at once.
//
// The wave-per-graph kernels spend ~1 scalar + ~1 vector instruction slot per walk step per graph and sit at
// the scalar-issue floor (profiles/r01/sq_counters_progress.md); a molecule has ~25 nodes, so 63 of 64 lanes
// idle in every vector instruction.  Here every lane runs the whole walk of its own graph — the same spec and
// token stream, bit-exact against oracle/gtok_oracle.c — so one vector instruction advances 64
// graphs.
//
// Requires GTOK_CSR_SIMPLE_SYMMETRIC (host-verified: no duplicate entry, every (u,v) has its (v,u) — any
// PyG-coalesced undirected graph): then the staged CSR rows ARE the adjacency lists, no bit matrix is built,
// and the edge type of a neighbour sits next to its id.  LDS per wave:
//   staged CSR chunk of the wave's 64 consecutive graphs (coalesced loads; u8 row pointers / neighbour ids /
//   types), indexed per lane by its own offsets;
//   vidx, rem  u8[maxn][64] laid out [index][lane] (lane l always hits bank group l: conflict-free for 64
//   different indices).  vidx = node -> visit index; rem[u] = unvisited neighbours left, which keeps `live` —
//   the visited nodes that still own an uncovered edge — incremental, so a dead end costs O(1).  Bracket
//   members are recovered from the row itself (the entry whose neighbour carries visit index k), so no
//   visit-order array is kept and the member's edge type comes with the entry.
// Tokens are stored straight to the row in HBM (lane-private, sequential); the pad tails of the 64 rows are
// filled cooperatively (coalesced) at the end.  Limits: maxn <= 64, maxe <= 255 (u8 indices).
#pragma once
#include "gtok_sent_reg.hpp"

namespace gtok {

struct alignas(4) Tok3 { int a, b, c; };   // token groups stored with one dword-aligned 12- / 8-byte write
struct alignas(4) Tok2 { int a, b; };

// k-th (0-based) set bit of w, per lane, k < popcount(w): branch-free halving on popcounts.  (A clear-lowest-bit
// loop runs max-over-lanes(k) times for the whole wave; picks from `live` have k up to the molecule's size.)
__device__ __forceinline__ int kth_bit64(uint64_t w, int k) {
  uint32_t x = (uint32_t)w;
  int base = 0, c = __popc(x);
  if (k >= c) { k -= c; x = (uint32_t)(w >> 32); base = 32; }
  c = __popc(x & 0xFFFFu); if (k >= c) { k -= c; x >>= 16; base += 16; }
  c = __popc(x & 0xFFu);   if (k >= c) { k -= c; x >>= 8;  base += 8; }
  c = __popc(x & 0xFu);    if (k >= c) { k -= c; x >>= 4;  base += 4; }
  c = __popc(x & 0x3u);    if (k >= c) { k -= c; x >>= 2;  base += 2; }
  return base + ((k >= (int)(x & 1u)) ? 1 : 0);
}

struct SentLaneArgs {
  gtok_csr g;
  gtok_sent_params p;
  int off_rp, off_col, off_eat, off_nat;   // staged CSR chunk of the wave (u8): row pointers, neighbour ids, edge / node types
  int off_vidx, off_rem;                   // u8 [maxn][64]: node -> visit index, unvisited neighbours left
  int lds;                                 // bytes of LDS per 64-graph wave
  int maxn;                                // rows of vidx / rem
  int cap_r, cap_n, cap_e;                 // staging capacities: row pointers, nodes, entries of one 64-graph chunk
  int32_t *out;
  int ld;
  int32_t *out_len;
  int units;                               // 64-graph units in the batch
  int *queue;                              // ticket counter block (gtok_common.hpp: Tickets)
};

template <bool LAB>
__global__ void __launch_bounds__(64, 2) sent_lane_kernel(const SentLaneArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  uint8_t *vidx = smem + a.off_vidx, *rem = smem + a.off_rem;
  uint8_t *srp = smem + a.off_rp, *scol = smem + a.off_col, *seat = smem + a.off_eat, *snat = smem + a.off_nat;
#define AT(arr, i) (arr)[(i) * 64 + lane]

  const int lim = a.p.max_len, ld = a.ld, cap = min(lim, ld);
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  const int node_off = idx_off + a.p.max_num_nodes;
  const int edge_off = node_off + a.p.num_node_types;
  const uint32_t k0 = (uint32_t)a.p.seed, k1 = (uint32_t)(a.p.seed >> 32), epoch = (uint32_t)a.p.epoch;
  const bool remap = a.p.remap_zinc != 0;   // folded into the emission constants (host guarantees maxn <= max_num_nodes)
  const int pos_base = remap ? 22 : idx_off;
  const int T_RESET = remap ? 2 : GTOK_SENT_RESET, T_LADJ = remap ? 2 : GTOK_SENT_LADJ;
  const int T_RADJ = remap ? 2 : GTOK_SENT_RADJ, T_EOS = remap ? 1 : GTOK_SENT_EOS;
  const int G = a.g.num_graphs;
  const int cap_r = a.cap_r, cap_e = a.cap_e, cap_n = a.cap_n;   // staging capacities

  // Units are drawn dynamically (a unit's time is set by its longest walk), and the loop is software-pipelined:
  // the loads of the NEXT unit's CSR chunk are issued before the CURRENT unit's padding stores.  vmcnt retires in
  // order on gfx9-family parts, so loads issued behind 40 KB of stores would wait for HBM to take them; issued
  // ahead, they return while the stores drain, and the stores drain behind the next walk (which loads nothing).
  // Staging moves 16 bytes per lane and load (4 column ids / 4 row pointers / 16 type bytes); the first passes of
  // all four arrays are in flight together.
  const bool lane0 = lane == 0;
  Tickets tickets;
  tickets.init(a.queue, (int)blockIdx.x, (int)gridDim.x, a.units);
  constexpr int UR = 8, UC = 16, UE = 4, UN = 2;   // 16-byte vectors per lane in flight: row pointers, column ids, types
  I32x4 rv[UR], cv[UC];
  U8x16 ev[UE], nv[UN];
  // the staged unit: group bounds (uniform) and this lane's graph
  int g0 = 0, gl = 0, N0 = 0, N1 = 0, nb0 = 0, nfull = 0, n = 0, e = 0;
  int64_t E0 = 0, E1 = 0, e0 = 0;
  bool valid = false;
#define GTOK_LANE_ISSUE(UNIT, MORE)                                                                               \
  {                                                                                                               \
    /* no next unit: every register is still (re)defined - a conditional issue would keep the old vectors live   \
       through the whole walk - from unit 0's addresses, with all counts zero */                                  \
    g0 = (MORE) ? (UNIT) * 64 : 0;                                                                                \
    gl = (MORE) ? min(g0 + 64, G) : 0;                                                                            \
    N0 = sload(a.g.node_ptr, g0); N1 = sload(a.g.node_ptr, gl);                                                   \
    E0 = sload(a.g.edge_ptr, g0); E1 = sload(a.g.edge_ptr, gl);                                                   \
    valid = g0 + lane < G;                                                                                        \
    nb0 = N0; nfull = 0; e0 = E0; e = 0;                                                                          \
    if (valid) {                                                                                                  \
      nb0 = a.g.node_ptr[g0 + lane];                                                                              \
      nfull = a.g.node_ptr[g0 + lane + 1] - nb0;                                                                  \
      e0 = a.g.edge_ptr[g0 + lane];                                                                               \
      e = min((int)(a.g.edge_ptr[g0 + lane + 1] - e0), a.g.max_edges);                                            \
    }                                                                                                             \
    n = min(nfull, a.maxn);                                                                                       \
    const int nrv_ = min((N1 - N0) + (gl - g0), cap_r) >> 2, ncv_ = (int)min(E1 - E0, (int64_t)cap_e) >> 2;       \
    const int nev_ = LAB ? ncv_ >> 2 : 0, nnv_ = LAB ? min(N1 - N0, cap_n) >> 4 : 0;                              \
    const I32x4 *rpv_ = reinterpret_cast<const I32x4 *>(a.g.rowptr + N0 + g0);                                    \
    const I32x4 *ccv_ = reinterpret_cast<const I32x4 *>(a.g.col + E0);                                            \
    const U8x16 *ecv_ = reinterpret_cast<const U8x16 *>(LAB ? a.g.eattr + E0 : nullptr);                          \
    const U8x16 *ncv2_ = reinterpret_cast<const U8x16 *>(LAB ? a.g.nattr + N0 : nullptr);                         \
    _Pragma("unroll") for (int j = 0; j < UR; ++j) { const int t = lane + j * kWave; rv[j] = t < nrv_ ? rpv_[t] : I32x4{0, 0, 0, 0}; } \
    if (LAB) {                                                                                                    \
      _Pragma("unroll") for (int j = 0; j < UE; ++j) { const int t = lane + j * kWave; ev[j] = t < nev_ ? ecv_[t] : U8x16{0, 0, 0, 0}; } \
      _Pragma("unroll") for (int j = 0; j < UN; ++j) { const int t = lane + j * kWave; nv[j] = t < nnv_ ? ncv2_[t] : U8x16{0, 0, 0, 0}; } \
    }                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < UC; ++j) { const int t = lane + j * kWave; cv[j] = t < ncv_ ? ccv_[t] : I32x4{0, 0, 0, 0}; } \
  }

  int unit = (int)blockIdx.x;
  GTOK_LANE_ISSUE(unit, unit < a.units);
  while (unit < a.units) {
    const int ticket = tickets.draw(lane0);
#ifdef GTOK_PHASE_TIMING   // profiling build only: cycle stamps per phase, left in the last columns of the unit's first row
    const uint64_t ts0 = __builtin_amdgcn_s_memtime();
#endif
    const int g = g0 + lane;
    // ---- commit the staged chunk to LDS (the previous unit's walk has finished reading it)
    __builtin_amdgcn_wave_barrier();
    {
      const int cr = min((N1 - N0) + (gl - g0), cap_r);   // graph g's row pointers start at node_ptr[g] + g
      const int ce = (int)min(E1 - E0, (int64_t)cap_e);
      const int cn = min(N1 - N0, cap_n);
      const int32_t *__restrict__ rpc = a.g.rowptr + N0 + g0;
      const int32_t *__restrict__ cc = a.g.col + E0;
      const uint8_t *__restrict__ ec = LAB ? a.g.eattr + E0 : nullptr;
      const uint8_t *__restrict__ nc = LAB ? a.g.nattr + N0 : nullptr;
      const I32x4 *rpv = reinterpret_cast<const I32x4 *>(rpc), *ccv = reinterpret_cast<const I32x4 *>(cc);
      const U8x16 *ecv = reinterpret_cast<const U8x16 *>(ec), *ncv = reinterpret_cast<const U8x16 *>(nc);
      uint32_t *srp4 = reinterpret_cast<uint32_t *>(srp), *scol4 = reinterpret_cast<uint32_t *>(scol);
      U8x16a *seat16 = reinterpret_cast<U8x16a *>(seat), *snat16 = reinterpret_cast<U8x16a *>(snat);
      const int nrv = cr >> 2, ncv4 = ce >> 2, nev = LAB ? ce >> 4 : 0, nnv = LAB ? cn >> 4 : 0;
      auto pack4 = [](const I32x4 &v) -> uint32_t {
        return ((uint32_t)v.x & 255u) | (((uint32_t)v.y & 255u) << 8) | (((uint32_t)v.z & 255u) << 16) | ((uint32_t)v.w << 24);
      };
#pragma unroll
      for (int j = 0; j < UR; ++j) { const int t = lane + j * kWave; if (t < nrv) srp4[t] = pack4(rv[j]); }
      if (LAB) {
#pragma unroll
        for (int j = 0; j < UE; ++j) { const int t = lane + j * kWave; if (t < nev) seat16[t] = U8x16a{ev[j].a, ev[j].b, ev[j].c, ev[j].d}; }
#pragma unroll
        for (int j = 0; j < UN; ++j) { const int t = lane + j * kWave; if (t < nnv) snat16[t] = U8x16a{nv[j].a, nv[j].b, nv[j].c, nv[j].d}; }
      }
#pragma unroll
      for (int j = 0; j < UC; ++j) { const int t = lane + j * kWave; if (t < ncv4) scol4[t] = pack4(cv[j]); }
      // chunks longer than the vectors in flight (not molecules), then the last < 4 / < 16 elements of each array
      for (int t = lane + UR * kWave; t < nrv; t += kWave) srp4[t] = pack4(rpv[t]);
      for (int t = lane + UC * kWave; t < ncv4; t += kWave) scol4[t] = pack4(ccv[t]);
      for (int t = lane + UE * kWave; t < nev; t += kWave) { const U8x16 x = ecv[t]; seat16[t] = U8x16a{x.a, x.b, x.c, x.d}; }
      for (int t = lane + UN * kWave; t < nnv; t += kWave) { const U8x16 x = ncv[t]; snat16[t] = U8x16a{x.a, x.b, x.c, x.d}; }
      if (lane < (cr & 3)) srp[(nrv << 2) + lane] = (uint8_t)rpc[(nrv << 2) + lane];
      if (lane < (ce & 3)) scol[(ncv4 << 2) + lane] = (uint8_t)cc[(ncv4 << 2) + lane];
      if (LAB && lane < (ce & 15)) seat[(nev << 4) + lane] = ec[(nev << 4) + lane];
      if (LAB && lane < (cn & 15)) snat[(nnv << 4) + lane] = nc[(nnv << 4) + lane];
    }
    wave_sync();
#ifdef GTOK_PHASE_TIMING
    const uint64_t ts1 = __builtin_amdgcn_s_memtime();
#endif
    const uint8_t *rpl = srp + (nb0 - N0) + lane;          // this lane's row pointers, neighbour ids, types
    const uint8_t *cl = scol + (int)(e0 - E0), *el = seat + (int)(e0 - E0), *nl = snat + (nb0 - N0);
    // A node's row: bounds + its first four neighbour ids in registers (one LDS round trip each); molecules
    // never have more, longer rows continue in a scalar tail loop.
    struct Row { int rs, re, u0, u1, u2, u3; };
    auto load_row = [&](int v) __attribute__((always_inline)) -> Row {
      Row r;
      r.rs = rpl[v];
      r.re = min((int)rpl[v + 1], e);
      const int last = max(r.re - 1, r.rs);
      r.u0 = cl[min(r.rs + 0, last)]; r.u1 = cl[min(r.rs + 1, last)];
      r.u2 = cl[min(r.rs + 2, last)]; r.u3 = cl[min(r.rs + 3, last)];
      return r;
    };
    auto row_mask = [&](const Row &r) __attribute__((always_inline)) -> uint64_t {
      const int deg = r.re - r.rs;
      uint64_t m = 0;
      m |= deg > 0 ? 1ull << r.u0 : 0ull; m |= deg > 1 ? 1ull << r.u1 : 0ull;
      m |= deg > 2 ? 1ull << r.u2 : 0ull; m |= deg > 3 ? 1ull << r.u3 : 0ull;
      for (int k = r.rs + 4; k < r.re; ++k) m |= 1ull << cl[k];
      return m;
    };
    // position of neighbour y inside the row (it is listed: symmetric adjacency)
    auto entry_of = [&](const Row &r, int y) __attribute__((always_inline)) -> int {
      const int deg = r.re - r.rs;
      int k = r.rs;
      k = (deg > 3 && r.u3 == y) ? r.rs + 3 : k;
      k = (deg > 2 && r.u2 == y) ? r.rs + 2 : k;
      k = (deg > 1 && r.u1 == y) ? r.rs + 1 : k;
      k = (deg > 0 && r.u0 == y) ? r.rs + 0 : k;
      for (int t = r.rs + 4; t < r.re; ++t) if (cl[t] == (uint8_t)y) k = t;
      return k;
    };
    // GTOK_CSR_SIMPLE_SYMMETRIC: no self-loops, no duplicates -> a node's unvisited-neighbour count starts at its degree
    for (int u = 0; u < n; ++u) AT(rem, u) = (uint8_t)max(min((int)rpl[u + 1], e) - (int)rpl[u], 0);

#ifdef GTOK_PHASE_TIMING
    const uint64_t ts2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- walk (per lane; mirrors oracle_sent step for step)
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);
    int32_t *__restrict__ orow = a.out + (int64_t)g * ld;
    uint64_t vis = 0, live = 0;
    int nvis = 0, pos = 0, d = 0, cur = 0;
    uint64_t plo = 0, phi = 0;

    auto emit = [&](int t) __attribute__((always_inline)) {
      if (pos < cap) orow[pos] = t;
      ++pos;
    };
    auto emit2 = [&](int t0, int t1) __attribute__((always_inline)) {   // two tokens, one dword-aligned 8-byte store
      if (pos + 1 < cap) { *reinterpret_cast<Tok2 *>(orow + pos) = Tok2{t0, t1}; pos += 2; }
      else { emit(t0); emit(t1); }
    };
    // decision d uses word d&3 of Philox block d>>2; the block is kept as two packed 64-bit values and the word
    // is extracted with mask arithmetic (a select chain over the captured words makes the compiler select
    // ADDRESSES and park them in scratch: two memory round trips per draw)
    auto below = [&](uint32_t nchoices) __attribute__((always_inline)) -> uint32_t {
      const int w = d & 3;
      if (w == 0) {
        uint32_t o[4];
        philox4x32_10((uint32_t)(d >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
        plo = ((uint64_t)o[1] << 32) | o[0];
        phi = ((uint64_t)o[3] << 32) | o[2];
      }
      const uint64_t m = 0ull - (uint64_t)((w >> 1) & 1);
      const uint32_t x = (uint32_t)(((plo & ~m) | (phi & m)) >> ((w & 1) << 5));
      ++d;
      return __umulhi(x, nchoices);
    };
    auto edge_tok = [&](int at) __attribute__((always_inline)) -> int {
      return remap ? remap_edge_type(at, edge_off) : edge_off + at;
    };
    // first visit of v; pred >= 0: reached over the trail edge (pred, v) whose edge-type token is `et`.
    // Returns v's row: the next step starts from it.
    auto visit = [&](int v, int pred, int et) __attribute__((always_inline)) -> Row {
      const int my = nvis;
      const Row r = load_row(v);
      const int deg = r.re - r.rs;
      // the four leading neighbours: their rem / vidx reads go out together
      const int u[4] = {r.u0, r.u1, r.u2, r.u3};
      int rm[4], vx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { rm[j] = AT(rem, u[j]); vx[j] = AT(vidx, u[j]); }
      uint64_t M = 0;   // bracket members, as bits in VISIT-INDEX space (ascending order for free)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < deg) {
          AT(rem, u[j]) = (uint8_t)(rm[j] - 1);           // one unvisited neighbour fewer for u
          if (rm[j] == 1) live &= ~(1ull << u[j]);
          if (((vis >> u[j]) & 1ull) && u[j] != pred) M |= 1ull << vx[j];
        }
      }
      for (int k = r.rs + 4; k < r.re; ++k) {             // long rows: scalar tail
        const int w = cl[k];
        const int q = (int)AT(rem, w) - 1;
        AT(rem, w) = (uint8_t)q;
        if (q == 0) live &= ~(1ull << w);
        if (((vis >> w) & 1ull) && w != pred) M |= 1ull << AT(vidx, w);
      }
      const uint64_t vis_before = vis;
      vis |= 1ull << v;
      AT(vidx, v) = (uint8_t)my;
      if (AT(rem, v) > 0) live |= 1ull << v;
      ++nvis;
      if (LAB) {   // [edge type] position type: one 12- / 8-byte store (dword-aligned) instead of 3 / 2 scattered ones
        const int x = nl[v];
        const int ty = remap ? remap_node_type(x, node_off, a.p.num_node_types) : node_off + x;
        if (pred >= 0) {
          if (pos + 2 < cap) { *reinterpret_cast<Tok3 *>(orow + pos) = Tok3{et, pos_base + my, ty}; pos += 3; }
          else { emit(et); emit(pos_base + my); emit(ty); }
        } else {
          if (pos + 1 < cap) { *reinterpret_cast<Tok2 *>(orow + pos) = Tok2{pos_base + my, ty}; pos += 2; }
          else { emit(pos_base + my); emit(ty); }
        }
      } else {
        emit(pos_base + my);
      }
      if (M) {
        emit(T_LADJ);
        while (M) {   // ascending visit index; the member with index k is the row entry whose neighbour carries it
          const int k = __builtin_ctzll(M);
          M &= M - 1;
          int kk = r.rs;
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const bool hit = j < deg && ((vis_before >> u[j]) & 1ull) && vx[j] == k && u[j] != pred;
            kk = hit ? r.rs + j : kk;
          }
          for (int t = r.rs + 4; t < r.re; ++t) {
            const int w = cl[t];
            if (((vis_before >> w) & 1ull) && AT(vidx, w) == k && w != pred) kk = t;
          }
          if (LAB) emit2(edge_tok(el[kk]), pos_base + k);
          else emit(pos_base + k);
        }
        emit(T_RADJ);
      }
      return r;
    };

    if (valid) {
      emit(GTOK_SENT_SOS);
      if (n > 0) {
        Row rc{0, 0, 0, 0, 0, 0};   // row of cur, carried from step to step (empty before the first visit)
        const uint64_t nodes = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        // Every step draws exactly one decision, so the draw, the k-th-member pick and the visit are written ONCE
        // and the step's kind only selects their operands.  (With one copy per kind, a wave whose lanes are in
        // different kinds - nearly every step - ran the Philox refill, the pick and the visit once per kind.)
        while (pos < lim) {
          const uint64_t row = row_mask(rc) & ~vis;
          // 0: extend the trail over an uncovered edge (always towards an unvisited node); 1: dead end, restart from
          // a visited node that still owns uncovered edges; 2: another component or an isolated node
          const int kind = row ? 0 : (live ? 1 : 2);
          if (kind == 2 && nvis >= n) break;
          const uint64_t set = kind == 0 ? row : (kind == 1 ? live : (~vis & nodes));
          const int pick = kth_bit64(set, (int)below((uint32_t)__popcll(set)));
          if (kind == 1) {
            emit2(T_RESET, pos_base + AT(vidx, pick));
            rc = load_row(pick);
          } else {
            int et = 0;
            if (kind == 0) { if (LAB) et = edge_tok(el[entry_of(rc, pick)]); }   // type of the listed entry cur->pick
            else if (nvis > 0) emit(T_RESET);   // (the walk's first node is a component start without RESET)
            rc = visit(pick, kind == 0 ? cur : -1, et);
          }
          cur = pick;
        }
      }
      emit(T_EOS);
    }
    int len = min(pos, lim);
    if (valid && a.p.query) {  // trainer/train_agtt.py:257-267: after the trail, original node ids, not remapped
      if (len + 0 < ld) orow[len + 0] = idx_off + nfull;
      if (len + 1 < ld) orow[len + 1] = idx_off + a.p.query[2 * (int64_t)g];
      if (len + 2 < ld) orow[len + 2] = idx_off + a.p.query[2 * (int64_t)g + 1];
      len += 3;
    }
    if (valid) a.out_len[g] = len;

#ifdef GTOK_PHASE_TIMING
    const uint64_t ts3 = __builtin_amdgcn_s_memtime();
#endif
    // ---- next unit: its loads go out ahead of this unit's padding stores
    const int lw = min(len, ld), done_g0 = g0;
    unit = tickets.settle(ticket, lane0);
    GTOK_LANE_ISSUE(unit, unit < a.units);
    // ---- pad the tails of the finished unit's 64 rows: four rows per pass, 16 lanes x 16-byte stores on each.
    // (Filling the unit's whole slab region up front was tried: every wave of a round then writes 51 KB at the
    // same moment and stalls ~30 us behind HBM; row-at-a-time 4-byte stores cost 27k cycles of loop overhead.)
    {
      const int q = lane & 15, pad = a.p.pad_id;
      for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + (lane >> 4);
        const int lr = __builtin_amdgcn_ds_bpermute(r << 2, lw);
        if (done_g0 + it * 4 >= G) break;
        if (done_g0 + r < G) {
          int32_t *__restrict__ row = a.out + (int64_t)(done_g0 + r) * ld + lr;
          const int nrem = ld - lr, nvec = nrem >> 2;
          for (int t = q; t < nvec; t += 16) reinterpret_cast<I32x4 *>(row)[t] = I32x4{pad, pad, pad, pad};
          if (q < (nrem & 3)) row[(nvec << 2) + q] = pad;
        }
      }
    }
#ifdef GTOK_PHASE_TIMING
    if (lane0 && ld >= 8) {
      const uint64_t ts4 = __builtin_amdgcn_s_memtime();
      int32_t *row = a.out + (int64_t)done_g0 * ld + ld - 4;
      row[0] = (int32_t)(ts1 - ts0); row[1] = (int32_t)(ts2 - ts1); row[2] = (int32_t)(ts3 - ts2); row[3] = (int32_t)(ts4 - ts3);
    }
#endif
  }
#undef GTOK_LANE_ISSUE
  tickets.retire(lane0, lane, (int)gridDim.x);
#undef AT
}

}  // namespace gtok
