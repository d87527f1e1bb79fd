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

__device__ __forceinline__ int kth_bit_serial(uint64_t w, int k) {   // per lane; k is small (degree-bounded)
  while (k-- > 0) w &= w - 1;
  return __builtin_ctzll(w);
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
  int units, upb;                          // 64-graph units in the batch / per workgroup
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
    // ---- stage the wave's CSR chunk: coalesced, independent loads (4 in flight per lane)
    wave_sync();
    {
      const int cr = min((N1 - N0) + (gl - g0), cap_r);   // graph g's row pointers start at node_ptr[g] + g
      const int32_t *__restrict__ rpc = a.g.rowptr + N0 + g0;
      for (int i = lane; i < cr; i += 4 * kWave) {
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j * kWave < cr) ? rpc[i + j * kWave] : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i + j * kWave < cr) srp[i + j * kWave] = (uint8_t)v[j];
      }
      const int ce = (int)min(E1 - E0, (int64_t)cap_e);
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
        const int cn = min(N1 - N0, cap_n);
        for (int i = lane; i < cn; i += kWave) snat[i] = a.g.nattr[N0 + i];
      }
    }
    wave_sync();
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
    for (int u = 0; u < n; ++u) AT(rem, u) = (uint8_t)__popcll(row_mask(load_row(u)) & ~(1ull << u));

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
    // first visit of v; pred >= 0: reached over the trail edge (pred, v) whose edge-type token is `et`
    auto visit = [&](int v, int pred, int et) __attribute__((always_inline)) {
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
          if (u[j] == v) { M |= 1ull << my; }            // self loop: v lists itself, last (largest visit index)
          else {
            AT(rem, u[j]) = (uint8_t)(rm[j] - 1);         // one unvisited neighbour fewer for u
            if (rm[j] == 1) live &= ~(1ull << u[j]);
            if (((vis >> u[j]) & 1ull) && u[j] != pred) M |= 1ull << vx[j];
          }
        }
      }
      for (int k = r.rs + 4; k < r.re; ++k) {             // long rows: scalar tail
        const int w = cl[k];
        if (w == v) { M |= 1ull << my; continue; }
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
            const bool hit = j < deg && (u[j] == v ? k == my : (((vis_before >> u[j]) & 1ull) && vx[j] == k && u[j] != pred));
            kk = hit ? r.rs + j : kk;
          }
          for (int t = r.rs + 4; t < r.re; ++t) {
            const int w = cl[t];
            if (w == v ? k == my : (((vis_before >> w) & 1ull) && AT(vidx, w) == k && w != pred)) kk = t;
          }
          if (LAB) emit(edge_tok(el[kk]));
          emit(pos_base + k);
        }
        emit(T_RADJ);
      }
    };

    if (valid) {
      emit(GTOK_SENT_SOS);
      if (n > 0) {
        cur = (int)below((uint32_t)n);
        visit(cur, -1, 0);
        while (pos < lim) {
          const Row rc = load_row(cur);
          const uint64_t row = row_mask(rc) & ~vis;
          if (row) {   // extend the trail over an uncovered edge (always towards an unvisited node)
            const int nxt = kth_bit_serial(row, (int)below((uint32_t)__popcll(row)));
            const int et = LAB ? edge_tok(el[entry_of(rc, nxt)]) : 0;   // type of the listed entry cur->nxt
            visit(nxt, cur, et);
            cur = nxt;
          } else if (live) {   // dead end: restart from a visited node that still owns uncovered edges
            cur = kth_bit_serial(live, (int)below((uint32_t)__popcll(live)));
            emit(T_RESET);
            emit(pos_base + AT(vidx, cur));
          } else if (nvis < n) {   // another component or an isolated node
            const uint64_t un = ~vis & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
            cur = kth_bit_serial(un, (int)below((uint32_t)(n - nvis)));
            emit(T_RESET);
            visit(cur, -1, 0);
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
