// gtok_sent_lane.hpp — SENT walk, LANE per graph: one wavefront tokenizes 64 small graphs at once.
//
// The wave-per-graph kernels spend ~1 scalar + ~1 vector instruction slot per walk step per graph and sit at
// the scalar-issue floor (profiles/r01/sq_counters_progress.md); a molecule has ~25 nodes, so 63 of 64 lanes
// idle in every vector instruction.  Here every lane runs the whole walk of its own graph — the same spec and
// token stream, bit-exact against oracle/gtok_oracle.c — so one vector instruction advances 64
// graphs.  Per-graph state lives in LDS as arrays laid out [index][lane]: element (i, lane) sits in bank group
// `lane` whatever i is, so 64 lanes indexing with 64 different i never conflict.
//   adj   u64[maxn][64]  immutable symmetric adjacency rows        order/vidx u8[maxn][64]  visit order and inverse
//   rem   u8 [maxn][64]  unvisited neighbours left per node: keeps the set of visited nodes that still own an
//                        uncovered edge (`live`, a register) incremental, so a dead end costs O(1)
//   rp/col/eat/nat u8    (labelled) the graph's own CSR, for edge-type lookups
// Tokens are stored straight to the row in HBM (lane-private, sequential); the pad tail of the 64 rows is
// filled cooperatively (coalesced) at the end.  Limits: maxn <= 64, maxe <= 255 (u8 indices).
//
// STATUS (round 1): opt-in with GTOK_SENT_KERNEL=lane.  It needs 450 instruction slots per molecule against
// 2075 for sent_reg_kernel, but 26 KB (unlabelled) / 41 KB (labelled) of LDS per 64-graph wave leave 6 / 3
// waves per CU, and it is latency-bound there: ZINC-full unlabelled 0.375 ms (reg kernel 0.405), labelled
// 0.93 ms (reg kernel 0.52).  Getting the labelled variant under the register kernel needs an LDS diet (edge
// types in a rank-indexed table instead of the staged CSR) — next round.
#pragma once
#include "gtok_sent_reg.hpp"

namespace gtok {

__device__ __forceinline__ int kth_bit_serial(uint64_t w, int k) {   // per lane; k is small (degree-bounded)
  while (k-- > 0) w &= w - 1;
  return __builtin_ctzll(w);
}

template <bool LAB>
__global__ void __launch_bounds__(64, 2) sent_lane_kernel(const SentArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  uint64_t *adj = reinterpret_cast<uint64_t *>(smem + a.l.adj);
  uint8_t *order = smem + a.l.order, *vidx = smem + a.l.vidx, *rem = smem + a.l.vis;
  // the wave's 64 graphs are one contiguous CSR chunk: staged here with coalesced loads, indexed per lane by
  // its own offsets.  Unlabelled walks only need it for the build, so order/vidx/rem alias it (host layout).
  uint8_t *srp = smem + a.l.rp, *scol = smem + a.l.col, *seat = smem + a.l.eat, *snat = smem + a.l.nat;
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

  const int vb = virtual_block();
  const int u0 = vb * a.upb, u1 = min(a.units, u0 + a.upb);
  for (int unit = u0; unit < u1; ++unit) {
    const int g0 = unit * 64;
    const int g = g0 + lane;
    const bool valid = g < G;
    const int gl = min(g0 + 64, G);                        // one past the wave's last graph
    const int N0 = sload(a.g.node_ptr, g0), N1 = sload(a.g.node_ptr, gl);
    const int64_t E0 = sload(a.g.edge_ptr, g0), E1 = sload(a.g.edge_ptr, gl);
    int nb0 = N0, nfull = 0, n = 0, e = 0;
    int64_t e0 = E0;
    if (valid) {
      nb0 = a.g.node_ptr[g];
      nfull = a.g.node_ptr[g + 1] - nb0;
      n = min(nfull, a.maxn);
      e0 = a.g.edge_ptr[g];
      e = min((int)(a.g.edge_ptr[g + 1] - e0), a.g.max_edges);
    }
    // ---- stage the chunk (coalesced, independent loads), then each lane builds its adjacency from LDS
    wave_sync();   // the previous unit's order/vidx/rem may alias the staging area
    {
      const int cr = min((N1 - N0) + (gl - g0), (a.maxn + 1) * 64);   // row pointers: graph g's slice starts at node_ptr[g]+g
      const int32_t *__restrict__ rpc = a.g.rowptr + N0 + g0;
      for (int i = lane; i < cr; i += 4 * kWave) {
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j * kWave < cr) ? rpc[i + j * kWave] : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i + j * kWave < cr) srp[i + j * kWave] = (uint8_t)v[j];
      }
      const int ce = (int)min(E1 - E0, (int64_t)a.g.max_edges * 64);
      const int32_t *__restrict__ cc = a.g.col + E0;
      for (int i = lane; i < ce; i += 4 * kWave) {
        int v[4], t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = i + j * kWave < ce;
          v[j] = in ? cc[i + j * kWave] : 0;
          t[j] = (LAB && in) ? (int)a.g.eattr[E0 + i + j * kWave] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i + j * kWave < ce) { scol[i + j * kWave] = (uint8_t)v[j]; if (LAB) seat[i + j * kWave] = (uint8_t)t[j]; }
      }
      if (LAB) {
        const int cn = min(N1 - N0, a.maxn * 64);
        for (int i = lane; i < cn; i += kWave) snat[i] = a.g.nattr[N0 + i];
      }
    }
    for (int u = 0; u < n; ++u) AT(adj, u) = 0;
    wave_sync();
    const uint8_t *rpl = srp + (nb0 - N0) + lane;          // this lane's row pointers, neighbour ids, types
    const uint8_t *cl = scol + (int)(e0 - E0), *el = seat + (int)(e0 - E0), *nl = snat + (nb0 - N0);
    for (int u = 0; u < n; ++u) {
      const int rs = rpl[u], re = rpl[u + 1];
      for (int k = rs; k < re && k < e; ++k) {
        const int v = cl[k];
        if ((unsigned)v < (unsigned)n) { AT(adj, u) |= 1ull << v; AT(adj, v) |= 1ull << u; }
      }
    }
    if (!LAB) wave_sync();   // staging is dead from here on: order/vidx/rem reuse it
    for (int u = 0; u < n; ++u) AT(rem, u) = (uint8_t)__popcll(AT(adj, u) & ~(1ull << u));

    // ---- walk (per lane; mirrors oracle_sent step for step)
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);
    int32_t *__restrict__ orow = a.out + (int64_t)g * ld;
    uint64_t vis = 0, live = 0;
    int nvis = 0, pos = 0, d = 0, cur = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;

    auto emit = [&](int t) __attribute__((always_inline)) {
      if (pos < cap) orow[pos] = t;
      ++pos;
    };
    auto below = [&](uint32_t nchoices) __attribute__((always_inline)) -> uint32_t {
      const int w = d & 3;
      if (w == 0) {
        uint32_t o[4];
        philox4x32_10((uint32_t)(d >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
        o0 = o[0]; o1 = o[1]; o2 = o[2]; o3 = o[3];
      }
      uint32_t x = o3;
      x = w == 2 ? o2 : x;
      x = w == 1 ? o1 : x;
      x = w == 0 ? o0 : x;
      ++d;
      return __umulhi(x, nchoices);
    };
    // edge-type TOKEN of (x,y): first listed entry x->y, else first y->x
    auto edge_token = [&](int x, int y) __attribute__((always_inline)) -> int {
      int at = 0;
      bool found = false;
      for (int k = rpl[x], ke = rpl[x + 1]; k < ke && !found; ++k)
        if (cl[k] == (uint8_t)y) { at = el[k]; found = true; }
      for (int k = rpl[y], ke = rpl[y + 1]; k < ke && !found; ++k)
        if (cl[k] == (uint8_t)x) { at = el[k]; found = true; }
      return remap ? remap_edge_type(at, edge_off) : edge_off + at;
    };
    // first visit of v; pred >= 0: reached over the trail edge (pred, v)
    auto visit = [&](int v, int pred) __attribute__((always_inline)) {
      const int my = nvis;
      const uint64_t row = AT(adj, v);
      uint64_t nbm = row & ~(1ull << v), M = 0;
      while (nbm) {   // neighbours of v: one unvisited neighbour fewer each; visited ones (bar pred) join the bracket
        const int u = __builtin_ctzll(nbm);
        nbm &= nbm - 1;
        const int r = (int)AT(rem, u) - 1;
        AT(rem, u) = (uint8_t)r;
        if (r == 0) live &= ~(1ull << u);
        if (((vis >> u) & 1ull) && u != pred) M |= 1ull << AT(vidx, u);
      }
      if ((row >> v) & 1ull) M |= 1ull << my;   // self loop: v lists itself, last (largest visit index)
      vis |= 1ull << v;
      AT(order, my) = (uint8_t)v;
      AT(vidx, v) = (uint8_t)my;
      if (AT(rem, v) > 0) live |= 1ull << v;
      ++nvis;
      if (LAB && pred >= 0) emit(edge_token(pred, v));
      emit(pos_base + my);
      if (LAB) {
        const int x = nl[v];
        emit(remap ? remap_node_type(x, node_off, a.p.num_node_types) : node_off + x);
      }
      if (M) {
        emit(T_LADJ);
        while (M) {   // ascending visit index
          const int k = __builtin_ctzll(M);
          M &= M - 1;
          if (LAB) emit(edge_token(v, AT(order, k)));
          emit(pos_base + k);
        }
        emit(T_RADJ);
      }
    };

    if (valid) {
      emit(GTOK_SENT_SOS);
      if (n > 0) {
        cur = (int)below((uint32_t)n);
        visit(cur, -1);
        while (pos < lim) {
          const uint64_t row = AT(adj, cur) & ~vis;
          if (row) {   // extend the trail over an uncovered edge (always towards an unvisited node)
            const int nxt = kth_bit_serial(row, (int)below((uint32_t)__popcll(row)));
            visit(nxt, cur);
            cur = nxt;
          } else if (live) {   // dead end: restart from a visited node that still owns uncovered edges
            cur = kth_bit_serial(live, (int)below((uint32_t)__popcll(live)));
            emit(T_RESET);
            emit(pos_base + AT(vidx, cur));
          } else if (nvis < n) {   // another component or an isolated node
            const uint64_t un = ~vis & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
            cur = kth_bit_serial(un, (int)below((uint32_t)(n - nvis)));
            emit(T_RESET);
            visit(cur, -1);
          } else {
            break;
          }
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

    // ---- pad tails of the wave's 64 rows, coalesced
    const int lw = min(len, ld);
    for (int j = 0; j < 64; ++j) {
      if (g0 + j >= G) break;
      const int lj = __builtin_amdgcn_readlane(lw, j);
      int32_t *__restrict__ r = a.out + (int64_t)(g0 + j) * ld;
      for (int i = lj + lane; i < ld; i += kWave) r[i] = a.p.pad_id;
    }
  }
#undef AT
}

}  // namespace gtok
