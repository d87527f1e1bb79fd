// gtok_sent_lds.hpp — SENT walk for graphs of up to 64*W nodes (W = 1, 2, 4, 8): the adjacency lives in LDS as
// an IMMUTABLE bit matrix adj[n][W]; the only walk state is the visited set, because for a visited node c the
// uncovered edges are exactly adj[c] & ~vis, and a new node's neighbourhood bracket is adj[v] & vis minus the
// trail edge just taken.  W-word sets are LANE-DISTRIBUTED: lane l holds word l % W of the
// visited set and of every row it fetches, so a set operation is one VALU op for all W words, a cardinality
// is a popcount + a log2(W)-step DPP prefix, and only the one word a decision lands in moves to SGPRs
// (an earlier version kept all W words in SGPR pairs: ~150 scalar instructions per visit at W = 4).  Decisions come from a register-resident,
// decision-major Philox block (64 decisions per refill); tokens go to an LDS row with unpredicated, ordered
// all-lane stores (lane j -> slot pos+j, junk beyond the step's tokens is overwritten by later steps).
// Same spec and token stream as sent_reg_kernel; bit-exact checker oracle/gtok_oracle.c:oracle_sent.
#pragma once
#include "gtok_sent_reg.hpp"

namespace gtok {

template <int W, bool LAB>
__global__ void __launch_bounds__(256) sent_lds_kernel(const SentArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = wave_id(), wpb = (int)(blockDim.x >> 6);
  unsigned char *base = smem + (size_t)wave * a.l.stride;
  uint64_t *adj = reinterpret_cast<uint64_t *>(base + a.l.adj);
  uint16_t *order = reinterpret_cast<uint16_t *>(base + a.l.order);   // completed 64-blocks of the visit order
  uint16_t *tok = reinterpret_cast<uint16_t *>(base + a.l.tok);
  uint16_t *colL = reinterpret_cast<uint16_t *>(base + a.l.col);
  int32_t *rp = reinterpret_cast<int32_t *>(base + a.l.rp);     // labelled only
  uint8_t *eatL = base + a.l.eat;                                // labelled only
  uint8_t *natL = base + a.l.nat;                                // labelled only

  const int lim = a.p.max_len;
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  const int node_off = idx_off + a.p.max_num_nodes;
  const int edge_off = node_off + a.p.num_node_types;
  const uint32_t k0 = (uint32_t)a.p.seed, k1 = (uint32_t)(a.p.seed >> 32), epoch0 = (uint32_t)a.p.epoch;
  const bool remap = a.p.remap_zinc != 0;
  const bool u16 = (a.p.flags & GTOK_SENT_U16) != 0;               // rows of 16-bit ids (include/gtok.h)
  constexpr int per = LAB ? 2 : 1;
  const bool is0 = lane == 0, is1 = lane == 1;
#define GTOK_TOK_ORDER() asm volatile("" ::: "memory")

  // graphs are drawn one at a time (gtok_common.hpp: Tickets): 10..256-node graphs differ >10x in walk length
  // (K epochs in one launch: tickets run over the G x K (epoch, graph) pairs, epoch-major - pair gv is graph gv mod G in
  // epoch gv / G and row gv of the [K, G, ld] slab)
  const int G = a.g.num_graphs, K = a.epochs, GV = G * K;
  const int nwaves = (int)(gridDim.x * wpb), wave_index = (int)(blockIdx.x * wpb + wave);
  Tickets tickets;
  tickets.init(a.queue, wave_index, nwaves, GV);
  int gv = wave_index;
  while (gv < GV) {
    const int ticket = tickets.draw(is0);
    const int g = K > 1 ? gv % G : gv;
    const uint32_t epoch = epoch0 + (uint32_t)(K > 1 ? gv / G : 0);
#ifdef GTOK_PHASE_TIMING   // profiling build only: cycle stamps per phase, left in the row's last columns
    const uint64_t ts0 = __builtin_amdgcn_s_memtime();
#endif
    const int nb0 = sload(a.g.node_ptr, g);
    const int nfull = sload(a.g.node_ptr, g + 1) - nb0;
    const int n = min(nfull, a.maxn);
    const int64_t e0 = sload(a.g.edge_ptr, g);
    const int e = min((int)(sload(a.g.edge_ptr, g + 1) - e0), a.g.max_edges);
    const int32_t *__restrict__ rpg = a.g.rowptr + nb0 + g;
    const int32_t *__restrict__ colg = a.g.col + e0;
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);

    // ---- stage (coalesced) and build the symmetric closure
    for (int i = lane; i < n * W; i += kWave) adj[i] = 0;
    if (LAB) {   // labelled walks look edge types up in the neighbour lists: stage them (coalesced)
      for (int i = lane; i < e; i += kWave) {
        colL[i] = (uint16_t)colg[i];
        eatL[i] = a.g.eattr[e0 + i];
      }
      for (int i = lane; i <= n; i += kWave) rp[i] = rpg[i];
      for (int i = lane; i < n; i += kWave) natL[i] = a.g.nattr[nb0 + i];
    }
    auto draws = [&](int d0) -> uint32_t {   // lane j: word of decision d0 + j
      uint32_t o[4];
      philox4x32_10((uint32_t)((d0 + lane) >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
      const int w = lane & 3;
      uint32_t x = o[3];
      x = w == 2 ? o[2] : x;
      x = w == 1 ? o[1] : x;
      x = w == 0 ? o[0] : x;
      return x;
    };
    uint32_t R = draws(0);
    wave_sync();
#ifdef GTOK_PHASE_TIMING
    const uint64_t ts1 = __builtin_amdgcn_s_memtime();
#endif
    // lane = row.  Unlabelled batches read the neighbour ids straight from HBM: 16 loads per lane are in flight per
    // round trip (4 made the build - a chain of ~1 us round trips, rows of 10-40 entries - 16 % of the kernel's time;
    // the LDS atomics themselves are ~1000 cycles per graph).  All rows of a graph with <= 64 nodes go out together,
    // larger graphs in passes of 64 rows.
    constexpr int BU = LAB ? 4 : 16;
    for (int u = lane; u < n; u += kWave) {
      const int rs = rpg[u], re = rpg[u + 1];
      for (int k0e = rs; k0e < re; k0e += BU) {
        int v[BU];
#pragma unroll
        for (int j = 0; j < BU; ++j) v[j] = LAB ? (int)colL[min(k0e + j, re - 1)] : colg[min(k0e + j, re - 1)];
#pragma unroll
        for (int j = 0; j < BU; ++j) {
          if (k0e + j < re && (unsigned)v[j] < (unsigned)n) {
            atomicOr(reinterpret_cast<unsigned long long *>(&adj[u * W + (v[j] >> 6)]), 1ull << (v[j] & 63));
            atomicOr(reinterpret_cast<unsigned long long *>(&adj[v[j] * W + (u >> 6)]), 1ull << (u & 63));
          }
        }
      }
    }
    wave_sync();
#ifdef GTOK_PHASE_TIMING
    const uint64_t ts2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- walk
    constexpr uint64_t kGroup = W >= 64 ? ~0ull : ((1ull << W) - 1ull);   // lanes 0..W-1: one copy of a set
    const int wl = lane & (W - 1);               // the word of a W-word set this lane holds
    uint64_t vis = 0;                            // visited set, word wl
    int ord = 0;                                 // lane k: the node visited (64*(nvis >> 6) + k)-th; completed
                                                 // 64-blocks of the visit order move to order[] in LDS
    uint64_t curw = 0;                           // word wl of the set {cur}
    const int rem_n = n - wl * 64;
    const uint64_t validw = rem_n >= 64 ? ~0ull : (rem_n > 0 ? ((1ull << rem_n) - 1ull) : 0ull);
    // One scalar unit serves the CU's four SIMDs, and this kernel issues about as many scalar as vector instructions
    // (4.4 k vs 4.7 k per graph: the scalar pipe, not the vector pipes, is what fills up).  The token cursor only
    // feeds LDS addresses, so it lives in a VECTOR register (the same value in every lane, seeded from an opaque
    // zero so that the compiler does not move it back): its arithmetic leaves the scalar pipe.
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    int pos = vz + 1;
    int d = 0, nvis = 0, cur = 0;

    auto below = [&](uint32_t nchoices) -> uint32_t {
      const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)R, d & 63);
      ++d;
      if ((d & 63) == 0) {
        int d0 = d;
        asm volatile("" : "+s"(d0));
        R = draws(d0);
      }
      return __umulhi(x, nchoices);
    };
    // lanes 0..W-1: c[0] + ... + c[lane]  (row_shr DPP steps; lanes past W-1 hold junk nobody reads)
    auto prefix = [&](int c) -> int {
      int x = c;
      if (W > 1) x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
      if (W > 2) x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
      if (W > 4) x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
      return x;
    };
    auto total_of = [&](int incl) -> int { return __builtin_amdgcn_readlane(incl, W - 1); };
    // k-th member (ascending node id; k wave-uniform, below the set's size) of a distributed set with per-word
    // counts c and inclusive prefix incl
    auto member = [&](uint64_t set, int c, int incl, int k) -> int {
      if (W == 1) return kth_bit_reg(readlane64(set, 0), k);
      const int ws = __popcll((uint64_t)__ballot(incl <= k) & kGroup);
      const int kk = k - __builtin_amdgcn_readlane(incl - c, ws);
      return ws * 64 + kth_bit_reg(readlane64(set, ws), kk);
    };
    // edge type of (x,y), both wave-uniform: first listed entry x->y, else first y->x
    auto etype_uniform = [&](int x, int y) -> int {
      for (int pass = 0; pass < 2; ++pass) {
        const int r = pass ? y : x, c = pass ? x : y;
        const int rs = rp[r], re = rp[r + 1];
        for (int b0 = rs; b0 < re; b0 += kWave) {
          const int k = b0 + lane;
          const uint64_t m = __ballot((k < re) && (colL[k] == (uint16_t)c));
          if (m) return eatL[b0 + __builtin_ctzll(m)];
        }
      }
      return 0;
    };
    auto etype_lane = [&](int x, int y) -> int {   // x uniform, y per lane
      for (int k = rp[x], ke = rp[x + 1]; k < ke; ++k)
        if (colL[k] == (uint16_t)y) return eatL[k];
      for (int k = rp[y], ke = rp[y + 1]; k < ke; ++k)
        if (colL[k] == (uint16_t)x) return eatL[k];
      return 0;
    };
    // first visit of v; pred >= 0: reached over the trail edge (pred, v).  Returns v's adjacency row (word wl):
    // the next trail step needs exactly that row, so it is fetched once.
    auto visit = [&](int v, int pred) -> uint64_t {
      ord = (lane == (nvis & 63)) ? v : ord;
      const uint64_t predw = pred >= 0 ? curw : 0ull;   // callers pass pred = cur (or none)
      curw = (wl == (v >> 6)) ? (1ull << (v & 63)) : 0ull;   // v becomes cur
      vis |= curw;
      if (LAB) {
        const int ty = node_off + natL[v];
        if (pred >= 0) {
          const int et = edge_off + etype_uniform(pred, v);
          int t = ty;
          t = is1 ? idx_off + nvis : t;
          t = is0 ? et : t;
          tok[pos + lane] = (uint16_t)t;
          GTOK_TOK_ORDER();
          pos += 3;
        } else {
          tok[pos + lane] = (uint16_t)(is0 ? idx_off + nvis : ty);
          GTOK_TOK_ORDER();
          pos += 2;
        }
      } else {
        tok[pos + lane] = (uint16_t)(idx_off + nvis);
        GTOK_TOK_ORDER();
        pos += 1;
      }
      ++nvis;
      const int top = (nvis - 1) >> 6;               // the 64-block of the visit order still in `ord`
      if ((nvis & 63) == 0) order[top * 64 + lane] = (uint16_t)ord;   // block complete: park it
      const uint64_t rowv = adj[v * W + wl];
      // visited neighbours of v other than pred: none -> no bracket
      if (((uint64_t)__ballot((rowv & vis & ~predw) != 0) & kGroup) == 0) return rowv;
      int cnt = 0;
      for (int c = 0; c <= top; ++c) {   // lane = visit index: members in ascending visit order
        const int k = c * 64 + lane;
        const int nb = c == top ? ord : (int)order[k];   // (stale lanes of ord hold in-range node ids)
        const uint64_t word = adj[v * W + (nb >> 6)];
        const bool member = k < nvis && ((word >> (nb & 63)) & 1ull) && nb != pred;
        const uint64_t M = __ballot(member);
        if (member) {
          const int q = pos + 1 + per * (cnt + mbcnt64(M));
          if (LAB) { tok[q] = (uint16_t)(edge_off + etype_lane(v, nb)); tok[q + 1] = (uint16_t)(idx_off + k); }
          else tok[q] = (uint16_t)(idx_off + k);
        }
        cnt += __popcll(M);
      }
      if (is0) { tok[pos] = GTOK_SENT_LADJ; tok[pos + 1 + per * cnt] = GTOK_SENT_RADJ; }
      GTOK_TOK_ORDER();
      pos += 2 + per * cnt;
      return rowv;
    };

    if (is0) tok[0] = GTOK_SENT_SOS;
    if (n > 0) {
      cur = (int)below((uint32_t)n);
      uint64_t rowc = visit(cur, -1);   // adjacency row of cur, word wl
      while (uni(pos) < lim) {
        {   // extend the trail over an uncovered edge (always towards an unvisited node)
          const uint64_t cand = rowc & ~vis;
          const int c = __popcll(cand), incl = prefix(c), cnt = total_of(incl);
          if (cnt) {
            const int nxt = member(cand, c, incl, (int)below((uint32_t)cnt));
            rowc = visit(nxt, cur);
            cur = nxt;
            continue;
          }
        }
        // dead end: visited nodes that still own uncovered edges (lane = node; needs every vis word)
        uint64_t visS[W];
#pragma unroll
        for (int w = 0; w < W; ++w) visS[w] = readlane64(vis, w);
        uint64_t live = 0;
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const int v = c * 64 + lane;
          uint64_t anyw = 0;
          if (v < n) {
#pragma unroll
            for (int w = 0; w < W; ++w) anyw |= adj[v * W + w] & ~visS[w];
          }
          const uint64_t mc = (c * 64 < n) ? ((uint64_t)__ballot(anyw != 0) & visS[c]) : 0ull;
          live = (wl == c) ? mc : live;
        }
        const int lc = __popcll(live), lincl = prefix(lc), total = total_of(lincl);
        if (total) {
          cur = member(live, lc, lincl, (int)below((uint32_t)total));
          rowc = adj[cur * W + wl];
          curw = (wl == (cur >> 6)) ? (1ull << (cur & 63)) : 0ull;
          int kidx = 0;
          for (int c = 0, top = (nvis - 1) >> 6; c <= top; ++c) {   // its visit index
            const int nb = c == top ? ord : (int)order[c * 64 + lane];
            const uint64_t hit = __ballot(c * 64 + lane < nvis && nb == cur);
            if (hit) { kidx = c * 64 + __builtin_ctzll(hit); break; }
          }
          tok[pos + lane] = (uint16_t)(is0 ? GTOK_SENT_RESET : idx_off + kidx);
          GTOK_TOK_ORDER();
          pos += 2;
          continue;
        }
        if (nvis < n) {  // another component or an isolated node
          const uint64_t fresh = ~vis & validw;
          const int fc = __popcll(fresh);
          cur = member(fresh, fc, prefix(fc), (int)below((uint32_t)(n - nvis)));
          tok[pos + lane] = (uint16_t)GTOK_SENT_RESET;
          GTOK_TOK_ORDER();
          pos += 1;
          rowc = visit(cur, -1);
          continue;
        }
        break;
      }
    }
    tok[pos + lane] = (uint16_t)GTOK_SENT_EOS;
    GTOK_TOK_ORDER();
    pos += 1;

#ifdef GTOK_PHASE_TIMING
    const uint64_t ts3 = __builtin_amdgcn_s_memtime();
#endif
    // ---- row out
    const int ltrail = min(uni(pos), lim);
    int len = ltrail;
    if (a.p.query) {  // trainer/train_agtt.py:257-267: after the trail, original node ids, not remapped
      if (lane < 3)
        tok[ltrail + lane] = (uint16_t)(idx_off + (lane == 0 ? nfull : a.p.query[2 * (int64_t)g + lane - 1]));
      len = ltrail + 3;
    }
    wave_sync();
    auto final_id = [=](int t, int i) -> int { return (remap && i < ltrail) ? remap_zinc_token(t, idx_off, node_off, edge_off) : t; };
    if (u16) write_row_tok16(reinterpret_cast<uint16_t *>(a.out) + (int64_t)gv * a.ld, a.ld, min(len, a.ld), a.p.pad_id, tok, final_id);
    else write_row_tok16(a.out + (int64_t)gv * a.ld, a.ld, min(len, a.ld), a.p.pad_id, tok, final_id);
    if (is0) a.out_len[gv] = len;
    wave_sync();
#ifdef GTOK_PHASE_TIMING
    if (is0 && a.ld >= 8) {
      const uint64_t ts4 = __builtin_amdgcn_s_memtime();
      int32_t *row = a.out + (int64_t)gv * a.ld + a.ld - 4;
      row[0] = (int32_t)(ts1 - ts0); row[1] = (int32_t)(ts2 - ts1); row[2] = (int32_t)(ts3 - ts2); row[3] = (int32_t)(ts4 - ts3);
    }
#endif
    gv = tickets.settle(ticket, is0);
  }
  tickets.retire(is0, lane, nwaves);
#undef GTOK_TOK_ORDER
}

}  // namespace gtok
