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
#pragma once
#include "gtok_sent_reg.hpp"

namespace gtok {

struct __attribute__((aligned(4))) U32x2a4 { uint32_t lo, hi; };   // two dwords at a 4-byte aligned LDS address: ds_read2_b32

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
  int lds;                                 // bytes of LDS per 64-graph wave
  int maxn;                                // largest graph of the batch (<= 64)
  int cap_r, cap_n, cap_e;                 // staging capacities: row pointers, nodes, entries of one 64-graph chunk
  int32_t *out;
  int ld;
  int32_t *out_len;
  int units;                               // 64-graph units in the batch
};

template <bool LAB, int P>
__global__ void __launch_bounds__(64, 4) sent_lane_kernel(const SentLaneArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  uint8_t *srp = smem + a.off_rp, *scol = smem + a.off_col, *seat = smem + a.off_eat, *snat = smem + a.off_nat;

  const int lim = a.p.max_len, ld = a.ld, cap = min(lim, ld);
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  const int node_off = idx_off + a.p.max_num_nodes;
  const int edge_off = node_off + a.p.num_node_types;
  const uint32_t k0 = (uint32_t)a.p.seed, k1 = (uint32_t)(a.p.seed >> 32), epoch = (uint32_t)a.p.epoch;
  const bool remap = a.p.remap_zinc != 0;   // folded into the emission constants (host guarantees maxn <= max_num_nodes)
  const int pos_base = remap ? 22 : idx_off;
  const uint64_t T_RESET = remap ? 2 : GTOK_SENT_RESET, T_LADJ = remap ? 2 : GTOK_SENT_LADJ;
  const uint64_t T_RADJ = remap ? 2 : GTOK_SENT_RADJ, T_EOS = remap ? 1 : GTOK_SENT_EOS;
  const int G = a.g.num_graphs, pad = a.p.pad_id;
  const int cap_r = a.cap_r, cap_e = a.cap_e, cap_n = a.cap_n;

  // ---- staging.  A unit's chunk is contiguous in every CSR array; it is loaded with 16-byte vectors (4 ids / 4 row
  // pointers / 16 type bytes per lane and load), packed to bytes and written to LDS in two phases so that at most
  // ~64 staging registers are live: A = row pointers + types, B = neighbour ids.
  constexpr int UR = 8, UC = 16, UE = 4, UN = 2;   // vectors per lane held in registers (chunks beyond: tail loops)
  struct Hdr { int g0, gl, N0, N1; int64_t E0, E1; int nb0, nfull, n, e; int64_t e0; bool valid; };
  auto header = [&](int unit) __attribute__((always_inline)) -> Hdr {
    Hdr h;
    h.g0 = unit * 64; h.gl = min(h.g0 + 64, G);
    h.N0 = sload(a.g.node_ptr, h.g0); h.N1 = sload(a.g.node_ptr, h.gl);
    h.E0 = sload(a.g.edge_ptr, h.g0); h.E1 = sload(a.g.edge_ptr, h.gl);
    h.valid = h.g0 + lane < G;
    h.nb0 = h.N0; h.nfull = 0; h.e0 = h.E0; h.e = 0;
    if (h.valid) {
      h.nb0 = a.g.node_ptr[h.g0 + lane];
      h.nfull = a.g.node_ptr[h.g0 + lane + 1] - h.nb0;
      h.e0 = a.g.edge_ptr[h.g0 + lane];
      h.e = min((int)(a.g.edge_ptr[h.g0 + lane + 1] - h.e0), a.g.max_edges);
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
      const int nev = (int)min(h.E1 - h.E0, (int64_t)cap_e) >> 4, nnv = min(h.N1 - h.N0, cap_n) >> 4;
      const U8x16 *ecv = reinterpret_cast<const U8x16 *>(a.g.eattr + h.E0);
      const U8x16 *ncv = reinterpret_cast<const U8x16 *>(a.g.nattr + h.N0);
#pragma unroll
      for (int j = 0; j < UE; ++j) { const int t = lane + j * kWave; r.ev[j] = t < nev ? ecv[t] : U8x16{0, 0, 0, 0}; }
#pragma unroll
      for (int j = 0; j < UN; ++j) { const int t = lane + j * kWave; r.nv[j] = t < nnv ? ncv[t] : U8x16{0, 0, 0, 0}; }
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
    for (int t = lane + UR * kWave; t < nrv; t += kWave) srp4[t] = pack4(rpv[t]);
    if (lane < (cr & 3)) srp[(nrv << 2) + lane] = (uint8_t)rpc[(nrv << 2) + lane];
    if (LAB) {
      const int ce = (int)min(h.E1 - h.E0, (int64_t)cap_e), cn = min(h.N1 - h.N0, cap_n);
      const uint8_t *__restrict__ ec = a.g.eattr + h.E0, *__restrict__ nc = a.g.nattr + h.N0;
      const U8x16 *ecv = reinterpret_cast<const U8x16 *>(ec), *ncv = reinterpret_cast<const U8x16 *>(nc);
      U8x16a *seat16 = reinterpret_cast<U8x16a *>(seat), *snat16 = reinterpret_cast<U8x16a *>(snat);
      const int nev = ce >> 4, nnv = cn >> 4;
#pragma unroll
      for (int j = 0; j < UE; ++j) { const int t = lane + j * kWave; if (t < nev) seat16[t] = U8x16a{r.ev[j].a, r.ev[j].b, r.ev[j].c, r.ev[j].d}; }
#pragma unroll
      for (int j = 0; j < UN; ++j) { const int t = lane + j * kWave; if (t < nnv) snat16[t] = U8x16a{r.nv[j].a, r.nv[j].b, r.nv[j].c, r.nv[j].d}; }
      for (int t = lane + UE * kWave; t < nev; t += kWave) { const U8x16 x = ecv[t]; seat16[t] = U8x16a{x.a, x.b, x.c, x.d}; }
      for (int t = lane + UN * kWave; t < nnv; t += kWave) { const U8x16 x = ncv[t]; snat16[t] = U8x16a{x.a, x.b, x.c, x.d}; }
      if (lane < (ce & 15)) seat[(nev << 4) + lane] = ec[(nev << 4) + lane];
      if (lane < (cn & 15)) snat[(nnv << 4) + lane] = nc[(nnv << 4) + lane];
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
    for (int t = lane + UC * kWave; t < ncv; t += kWave) scol4[t] = pack4(ccv[t]);
    if (lane < (ce & 3)) scol[(ncv << 2) + lane] = (uint8_t)cc[(ncv << 2) + lane];
  };

  // Units are dealt round-robin (a unit's time is the longest of its 64 walks: they are all alike); with 16
  // resident waves per CU a ZINC-full launch gives every wave exactly one unit.
  const int stride = (int)gridDim.x;
  int unit = virtual_block();
  if (unit >= a.units) return;
  Hdr h = header(unit);
  {
    RegsA ra;
    load_a(h, ra);
    commit_a(h, ra);
    stage_b(h);
  }
  for (;;) {
    wave_sync();
    const int g = h.g0 + lane;
    const bool valid = h.valid;
    const int n = h.n, e = h.e;
    const int rbase = (h.nb0 - h.N0) + lane, cbase = (int)(h.e0 - h.E0), nbase = h.nb0 - h.N0;

    // A node's row: bounds, the set of its neighbours, and the first four neighbour ids / edge types as packed
    // bytes (molecules rarely have more; longer rows continue in byte loops).
    struct Row { uint64_t mask; uint32_t nb4, et4; int rs, deg; };
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
      uint64_t m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= j < r.deg ? 1ull << ((r.nb4 >> (8 * j)) & 63u) : 0ull;
      for (int k = 4; k < r.deg; ++k) m |= 1ull << (scol[o + k] & 63u);
      r.mask = m;
      return r;
    };
    // edge type of the listed entry row -> y (y is listed: symmetric adjacency).  Zero-byte search over the four
    // packed ids (the lowest flag of the classic (x - 0x01..) & ~x & 0x80.. test is exact), byte loop beyond.
    auto find_et = [&](const Row &r, int y) __attribute__((always_inline)) -> int {
      const uint32_t x = r.nb4 ^ ((uint32_t)y * 0x01010101u);
      uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
      z &= r.deg >= 4 ? 0xFFFFFFFFu : ((1u << (8 * r.deg)) - 1u);
      int et = 0;
      if (z) {
        et = (int)((r.et4 >> (__builtin_ctz(z) - 7)) & 255u);
      } else {
        const int o = cbase + r.rs;
        for (int k = 4; k < r.deg; ++k) if (scol[o + k] == (uint8_t)y) et = seat[o + k];
      }
      return et;
    };

    // ---- bit-sliced counters: c[p] bit u = bit p of (number of unvisited neighbours of u); starts at the degree
    uint64_t c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) c[p] = 0;
    {
      int lo = n > 0 ? (int)srp[rbase] : 0;
      for (int u = 0; u < n; ++u) {
        const int hi = min((int)srp[rbase + u + 1], e);
        const uint32_t dg = (uint32_t)max(hi - lo, 0);
        lo = hi;
#pragma unroll
        for (int p = 0; p < P; ++p) c[p] |= (uint64_t)((dg >> p) & 1u) << u;
      }
    }

    // ---- walk (per lane; mirrors oracle_sent step for step)
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);
    int32_t *__restrict__ orow = a.out + (int64_t)g * ld;
    uint64_t vis = 0, live = 0, wlo = 0;
    int nvis = 0, pos = 0, fl = 0, d = 0, cur = 0;
    uint64_t plo = 0, phi = 0;

    // token window: tokens fl .. pos-1 of the row sit in wlo, 16 bits each (pos - fl <= 3 between appends)
    auto flush = [&](uint64_t w) __attribute__((always_inline)) {
      const I32x4 v{(int)(w & 0xFFFFu), (int)((w >> 16) & 0xFFFFu), (int)((w >> 32) & 0xFFFFu), (int)(w >> 48)};
      if (fl + 4 <= cap) {
        *reinterpret_cast<I32x4 *>(orow + fl) = v;
      } else {                                   // the row's cut (max_len or a narrow slab) falls inside this group
        if (fl + 0 < cap) orow[fl + 0] = v.x;
        if (fl + 1 < cap) orow[fl + 1] = v.y;
        if (fl + 2 < cap) orow[fl + 2] = v.z;
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

    if (valid) {
      append((uint64_t)GTOK_SENT_SOS, 1);
      if (n > 0) {
        Row rc{0, 0, 0, 0, 0};   // row of cur, carried from step to step (empty before the first visit)
        const uint64_t nodes = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        // Every step draws exactly one decision and touches exactly one node, so the draw, the pick, the row load
        // and the token group are written ONCE and the step's kind only selects operands.  (With one copy per
        // kind, a wave whose lanes are in different kinds - nearly every step - runs every copy.)
        while (pos < lim) {
          const uint64_t row = rc.mask & ~vis;
          // 0: extend the trail over an uncovered edge (always towards an unvisited node); 1: dead end, restart from
          // a visited node that still owns uncovered edges; 2: another component or an isolated node
          const int kind = row ? 0 : (live ? 1 : 2);
          if (kind == 2 && nvis >= n) break;
          const uint64_t set = kind == 0 ? row : (kind == 1 ? live : (~vis & nodes));
          const int pick = kth_bit64(set, (int)below((uint32_t)__popcll(set)));
          int et = 0;
          if (LAB && kind == 0) et = find_et(rc, pick);       // type of the listed entry cur -> pick
          const Row rn = load_row(pick);
          const int xb = snat[nbase + pick];                   // node type (first visit) or visit index (kind 1)
          const bool first = kind != 1;
          const int my = nvis;
          if (first) snat[nbase + pick] = (uint8_t)my;         // from now on this byte is the node's visit index
          // ---- the step's token group: [edge type | RESET] position [node type]
          {
            const uint64_t tpos = (uint64_t)(pos_base + (first ? my : xb));
            uint64_t ta = T_RESET, val;
            bool has_a = kind == 1 || (kind == 2 && nvis > 0);   // (the walk's first node is a component start without RESET)
            int cnt = 1;
            if (LAB) {
              if (kind == 0) { ta = (uint64_t)(remap ? remap_edge_type(et, edge_off) : edge_off + et); has_a = true; }
              const uint64_t ty = (uint64_t)(remap ? remap_node_type(xb, node_off, a.p.num_node_types) : node_off + xb);
              val = first ? (tpos | (ty << 16)) : tpos;
              cnt += first;
            } else {
              val = tpos;
            }
            if (has_a) { val = ta | (val << 16); ++cnt; }
            append(val, cnt);
          }
          // ---- first visit: neighbours lose an unvisited neighbour; already visited neighbours other than the
          // trail's predecessor are this node's bracket
          const uint64_t S = first ? rn.mask : 0ull;
          uint64_t M = S & vis & ~(kind == 0 ? 1ull << cur : 0ull);
          {
            uint64_t b = S, nz = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) { const uint64_t t = c[p]; c[p] = t ^ b; b &= ~t; nz |= c[p]; }
            vis |= 1ull << pick;
            live = vis & nz;
          }
          nvis += first;
          if (M) {   // LADJ, members by ascending visit index ([edge type] position), RADJ
            bool head = true;
            do {
              uint64_t t = M;
              int bu = 0, bv = 256;
              do {
                const int u = __builtin_ctzll(t);
                t &= t - 1;
                const int vx = snat[nbase + u];
                if (vx < bv) { bv = vx; bu = u; }
              } while (t);
              M &= ~(1ull << bu);
              uint64_t val = (uint64_t)(pos_base + bv);
              int cnt = 1;
              if (LAB) {
                const int at = find_et(rn, bu);
                val = (uint64_t)(remap ? remap_edge_type(at, edge_off) : edge_off + at) | (val << 16);
                cnt = 2;
              }
              if (head) { val = T_LADJ | (val << 16); ++cnt; head = false; }
              if (!M) { val |= T_RADJ << (cnt << 4); ++cnt; }
              append(val, cnt);
            } while (M);
          }
          rc = rn;
          cur = pick;
        }
      }
      append(T_EOS, 1);
    }
    // ---- end of the row: what is still in the window, the query tail (trainer/train_agtt.py:257-267: after the
    // trail, original node ids, not remapped), and pad up to the next multiple of 4
    const int len = min(pos, lim);
    int tot = len;
    int q0 = 0, q1 = 0, q2 = 0;
    if (valid && a.p.query) {
      q0 = idx_off + h.nfull; q1 = idx_off + a.p.query[2 * (int64_t)g]; q2 = idx_off + a.p.query[2 * (int64_t)g + 1];
      tot = len + 3;
    }
    if (valid) {
      a.out_len[g] = tot;
      const int stop = min(ld, (tot + 3) & ~3);
      for (int i = min(fl, len); i < stop; ++i) {
        int v = pad;
        if (i < len) {
          if (i < fl) continue;                  // already written by a window flush
          v = (int)((wlo >> ((i - fl) << 4)) & 0xFFFFu);
        } else if (i < tot) {
          v = i == len ? q0 : (i == len + 1 ? q1 : q2);
        }
        orow[i] = v;
      }
    }

    // ---- next unit: the loads of its phase A go out ahead of this unit's padding stores
    const int lw = valid ? min(ld, (tot + 3) & ~3) : 0, done_g0 = h.g0;
    const int next = unit + stride;
    const bool more = next < a.units;
    RegsA ra;
    if (more) { h = header(next); load_a(h, ra); }
    // ---- pad the tails of the finished unit's 64 rows: four rows per pass, 16 lanes x 16-byte stores on each
    {
      const int q = lane & 15;
      for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + (lane >> 4);
        const int lr = __builtin_amdgcn_ds_bpermute(r << 2, lw);
        if (done_g0 + it * 4 >= G) break;
        if (done_g0 + r < G) {
          int32_t *__restrict__ rowp = a.out + (int64_t)(done_g0 + r) * ld + lr;
          const int nrem = ld - lr, nvec = nrem >> 2;
          for (int t = q; t < nvec; t += 16) reinterpret_cast<I32x4 *>(rowp)[t] = I32x4{pad, pad, pad, pad};
          if (q < (nrem & 3)) rowp[(nvec << 2) + q] = pad;
        }
      }
    }
    if (!more) break;
    unit = next;
    __builtin_amdgcn_wave_barrier();
    commit_a(h, ra);
    stage_b(h);
  }
}

}  // namespace gtok
