// gtok_sent.hip — SENT (Segmented Eulerian Neighbourhood Trail) walk, gfx950.
//
// Replaces autograph's Graph2TrailTokenizer.__call__ as invoked at
// trainer/train_agtt.py:250, optionally fused with remap_zinc_tokens
// (trainer/train_agtt.py:171-244) and the shortest_path query append
// (trainer/train_agtt.py:257-267).  The walk follows the spec frozen in
// DESIGN.md §SENT (upstream AutoGraph parity is unpinned); the
// bit-exact checker is oracle/gtok_oracle.c:oracle_sent.
//
// One wavefront per graph.  LDS slice of a wave:
//   adj   uint64[maxN][W]  remaining (uncovered) undirected edges, bit matrix
//   vis   uint64[W]        visited-node set
//   vidx  uint16[maxN]     node -> visit index       order uint16[maxN] inverse
//   rng   uint32[256]      one wave-wide Philox4x32-10 fill = 256 decisions
//   tok   uint16[cap]      the trail, before remap / padding
//   (labelled) rp int32[maxN+1], col uint16[maxE], eat uint8[maxE], nat uint8[maxN]
// Wave primitives: __ballot + __ffsll pick the k-th remaining edge / restart
// node; __ballot + __popcll rank the members of a neighbourhood bracket.
#include <cstdlib>

#include "gtok_common.hpp"
#include "gtok.h"
#include "gtok_sent_reg.hpp"

namespace gtok {

template <int W, bool LAB>
struct SentWave {
  uint64_t *adj, *vis;
  uint16_t *vidx, *order, *tok, *colL;
  uint32_t *rng;
  int32_t *rp;
  uint8_t *eatL, *natL;
  int lane, cap, lim, pos, d, nvis, n;  // cap: tokens stored, lim: max_len (walk stops there)
  int idx_off, node_off, edge_off;
  uint32_t k0, k1, epoch, gid_lo, gid_hi;

  __device__ __forceinline__ void emit(int t) {
    if (lane == 0 && pos < cap) tok[pos] = (uint16_t)t;
    ++pos;
  }
  __device__ __forceinline__ void fill_rng(uint32_t first_block) {
    uint32_t o[4];
    philox4x32_10(first_block + (uint32_t)lane, epoch, gid_lo, gid_hi, k0, k1, o);
    rng[4 * lane + 0] = o[0]; rng[4 * lane + 1] = o[1];
    rng[4 * lane + 2] = o[2]; rng[4 * lane + 3] = o[3];
  }
  // decision d uses word d&3 of Philox block d>>2
  __device__ __forceinline__ uint32_t below(uint32_t nchoices) {
    const uint32_t r = uni(rng[d & 255]);
    ++d;
    if ((d & 255) == 0) {
      wave_sync();
      fill_rng((uint32_t)d >> 2);
      wave_sync();
    }
    return __umulhi(r, nchoices);
  }
  __device__ __forceinline__ uint64_t row_word(int v, int w) const { return adj[v * W + w]; }

  // edge type of (a,b), both wave-uniform: first CSR entry a->b, else first b->a
  __device__ __forceinline__ int etype_uniform(int a, int b) const {
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass ? b : a, c = pass ? a : b;
      const int rs = rp[r], re = rp[r + 1];
      for (int base = rs; base < re; base += kWave) {
        const int k = base + lane;
        const bool hit = (k < re) && (colL[k] == (uint16_t)c);
        const uint64_t m = __ballot(hit);
        if (m) return eatL[base + __ffsll((unsigned long long)m) - 1];
      }
    }
    return 0;
  }
  // same, per lane (a uniform, b lane-varying)
  __device__ __forceinline__ int etype_lane(int a, int b) const {
    for (int k = rp[a], re = rp[a + 1]; k < re; ++k)
      if (colL[k] == (uint16_t)b) return eatL[k];
    for (int k = rp[b], re = rp[b + 1]; k < re; ++k)
      if (colL[k] == (uint16_t)a) return eatL[k];
    return 0;
  }

  // first visit of v: position token, type token, neighbourhood bracket
  __device__ __forceinline__ void visit_new(int v) {
    if (lane == 0) {
      vidx[v] = (uint16_t)nvis;
      order[nvis] = (uint16_t)v;
      vis[v >> 6] |= 1ull << (v & 63);
    }
    emit(idx_off + nvis);
    if (LAB) emit(node_off + natL[v]);
    ++nvis;
    wave_sync();
    bool any = false;
#pragma unroll
    for (int w = 0; w < W; ++w) any |= (row_word(v, w) & vis[w]) != 0;
    if (!uni((int)any)) return;
    emit(GTOK_SENT_LADJ);
    const int per = LAB ? 2 : 1;
    for (int base = 0; base < nvis; base += kWave) {
      const int k = base + lane;
      int a = 0;
      bool member = false;
      if (k < nvis) {
        a = order[k];
        member = (row_word(v, a >> 6) >> (a & 63)) & 1ull;
      }
      const uint64_t m = __ballot(member);
      if (m == 0) continue;
      if (member) {
        const int q = pos + per * __popcll(m & lanemask_lt());
        if (LAB) {
          const int et = etype_lane(v, a);
          if (q < cap) tok[q] = (uint16_t)(edge_off + et);
          if (q + 1 < cap) tok[q + 1] = (uint16_t)(idx_off + k);
        } else {
          if (q < cap) tok[q] = (uint16_t)(idx_off + k);
        }
        if (a != v) adj[a * W + (v >> 6)] &= ~(1ull << (v & 63));
      }
      pos += per * __popcll(m);
    }
    emit(GTOK_SENT_RADJ);
    wave_sync();
    if (lane < W) adj[v * W + lane] &= ~vis[lane];  // back edges (and a self loop) are now covered
    wave_sync();
  }

  __device__ __forceinline__ int pick(const uint64_t (&m)[W], int total) {
    int k = (int)below((uint32_t)total), res = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int c = __popcll(m[w]);
      if (res < 0) {
        if (k < c) res = w * 64 + kth_bit(m[w], k);
        else k -= c;
      }
    }
    return res;
  }

  __device__ void walk() {
    emit(GTOK_SENT_SOS);
    if (n > 0) {
      int cur = (int)below((uint32_t)n);
      visit_new(cur);
      while (pos < lim) {
        uint64_t m[W];
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) { m[w] = uni(row_word(cur, w)); cnt += __popcll(m[w]); }
        if (cnt > 0) {  // extend the trail over an uncovered edge (always to an unvisited node)
          const int nxt = pick(m, cnt);
          if (nxt < 0) break;
          if (lane == 0) {
            adj[cur * W + (nxt >> 6)] &= ~(1ull << (nxt & 63));
            adj[nxt * W + (cur >> 6)] &= ~(1ull << (cur & 63));
          }
          if (LAB) emit(edge_off + etype_uniform(cur, nxt));
          wave_sync();
          visit_new(nxt);
          cur = nxt;
          continue;
        }
        // dead end: visited nodes that still own uncovered edges
        int total = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int v = w * 64 + lane;
          bool live = false;
          if (v < n) {
            uint64_t any = 0;
#pragma unroll
            for (int x = 0; x < W; ++x) any |= row_word(v, x);
            live = (any != 0) && ((vis[w] >> lane) & 1ull);
          }
          m[w] = (w * 64 < n) ? (uint64_t)__ballot(live) : 0ull;
          total += __popcll(m[w]);
        }
        if (total > 0) {
          emit(GTOK_SENT_RESET);
          cur = pick(m, total);
          if (cur < 0) break;
          emit(idx_off + uni((int)vidx[cur]));
          continue;
        }
        if (nvis < n) {  // another component or an isolated node
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int rem = n - w * 64;
            const uint64_t valid = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
            m[w] = ~uni(vis[w]) & valid;
          }
          emit(GTOK_SENT_RESET);
          cur = pick(m, n - nvis);
          if (cur < 0) break;
          visit_new(cur);
          continue;
        }
        break;
      }
    }
    emit(GTOK_SENT_EOS);
  }
};

template <int W, bool LAB>
__global__ void __launch_bounds__(256) sent_kernel(const SentArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = wave_id(), wpb = (int)(blockDim.x >> 6);
  unsigned char *base = smem + (size_t)wave * a.l.stride;

  SentWave<W, LAB> s;
  s.adj = reinterpret_cast<uint64_t *>(base + a.l.adj);
  s.vis = reinterpret_cast<uint64_t *>(base + a.l.vis);
  s.vidx = reinterpret_cast<uint16_t *>(base + a.l.vidx);
  s.order = reinterpret_cast<uint16_t *>(base + a.l.order);
  s.rng = reinterpret_cast<uint32_t *>(base + a.l.rng);
  s.tok = reinterpret_cast<uint16_t *>(base + a.l.tok);
  s.rp = reinterpret_cast<int32_t *>(base + a.l.rp);
  s.colL = reinterpret_cast<uint16_t *>(base + a.l.col);
  s.eatL = base + a.l.eat;
  s.natL = base + a.l.nat;
  s.lane = lane;
  s.cap = a.cap;
  s.lim = a.p.max_len;
  s.idx_off = GTOK_SENT_IDX_OFFSET;
  s.node_off = s.idx_off + a.p.max_num_nodes;
  s.edge_off = s.node_off + a.p.num_node_types;
  s.k0 = (uint32_t)a.p.seed; s.k1 = (uint32_t)(a.p.seed >> 32);
  s.epoch = (uint32_t)a.p.epoch;

  const int vb = virtual_block();
  const int u0 = vb * a.upb, u1 = min(a.units, u0 + a.upb);
  for (int unit = u0; unit < u1; ++unit) {
    const int g = unit * wpb + wave;
    if (g >= a.g.num_graphs) break;
    const int nb0 = a.g.node_ptr[g];
    const int n = min(a.g.node_ptr[g + 1] - nb0, a.maxn);
    const int64_t e0 = a.g.edge_ptr[g];
    const int e = LAB ? min((int)(a.g.edge_ptr[g + 1] - e0), a.g.max_edges) : (int)(a.g.edge_ptr[g + 1] - e0);
    const int32_t *__restrict__ rpg = a.g.rowptr + nb0 + g;
    const int32_t *__restrict__ colg = a.g.col + e0;
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    s.n = n; s.pos = 0; s.d = 0; s.nvis = 0;
    s.gid_lo = (uint32_t)gid; s.gid_hi = (uint32_t)(gid >> 32);

    for (int i = lane; i < n * W; i += kWave) s.adj[i] = 0;
    if (lane < W) s.vis[lane] = 0;
    s.fill_rng(0);
    if (LAB) {
      for (int i = lane; i <= n; i += kWave) s.rp[i] = rpg[i];
      for (int i = lane; i < e; i += kWave) {
        s.colL[i] = (uint16_t)colg[i];
        s.eatL[i] = a.g.eattr[e0 + i];
      }
      for (int i = lane; i < n; i += kWave) s.natL[i] = a.g.nattr[nb0 + i];
    }
    wave_sync();
    // undirected=True: symmetric closure of the listed entries
    for (int u = lane; u < n; u += kWave) {
      const int rs = rpg[u], re = rpg[u + 1];
      for (int k = rs; k < re; ++k) {
        const int v = LAB ? (int)s.colL[k] : colg[k];
        if ((unsigned)v < (unsigned)n) {
          atomicOr(reinterpret_cast<unsigned long long *>(&s.adj[u * W + (v >> 6)]), 1ull << (v & 63));
          atomicOr(reinterpret_cast<unsigned long long *>(&s.adj[v * W + (u >> 6)]), 1ull << (u & 63));
        }
      }
    }
    wave_sync();

    s.walk();

    const int ltrail = min(s.pos, a.p.max_len);  // true trail length; tokens beyond `cap` were not stored
    int len = ltrail;
    if (a.p.query) {  // trainer/train_agtt.py:257-267, original node ids, after any EOS, not remapped
      if (lane < 3 && ltrail + lane < a.cap + 3)
        s.tok[ltrail + lane] = (uint16_t)(s.idx_off + (lane == 0 ? a.g.node_ptr[g + 1] - nb0
                                                                 : a.p.query[2 * (int64_t)g + lane - 1]));
      len = ltrail + 3;
    }
    wave_sync();
    const bool remap = a.p.remap_zinc != 0;
    const int io = s.idx_off, no = s.node_off, eo = s.edge_off;
    const uint16_t *tok = s.tok;
    write_row(a.out + (int64_t)g * a.ld, a.ld, min(len, a.ld), a.p.pad_id, [=](int i) -> int {
      const int t = tok[i];
      return (remap && i < ltrail) ? remap_zinc_token(t, io, no, eo) : t;
    });
    if (lane == 0) a.out_len[g] = len;
    wave_sync();
  }
}

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace gtok

using namespace gtok;

extern "C" int gtok_sent(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids,
                         int32_t ld, int32_t *out_len, void *stream) {
  if (!g || !p || !out_ids || !out_len || ld <= 0 || g->num_graphs < 0) return GTOK_E_INVAL;
  if (g->num_graphs == 0) return GTOK_OK;
  if (!g->node_ptr || !g->edge_ptr || !g->rowptr || (g->max_edges > 0 && !g->col)) return GTOK_E_INVAL;
  if (p->max_len < 0 || p->max_num_nodes < 0) return GTOK_E_INVAL;
  if (p->labeled && (!g->nattr || !g->eattr)) return GTOK_E_INVAL;
  if (g->max_nodes > GTOK_MAX_NODES) return GTOK_E_TOO_LARGE;
  if (p->labeled && (g->max_edges > 65535 * 4)) return GTOK_E_TOO_LARGE;
  // tokens are staged as uint16
  if ((int64_t)GTOK_SENT_IDX_OFFSET + p->max_num_nodes + p->num_node_types + 512 > 65535) return GTOK_E_TOO_LARGE;
  if (g->max_nodes + GTOK_SENT_IDX_OFFSET > 65535) return GTOK_E_TOO_LARGE;

  const int maxn = g->max_nodes > 0 ? g->max_nodes : 1;
  const int W = maxn <= 64 ? 1 : maxn <= 128 ? 2 : maxn <= 256 ? 4 : 8;
  const int cap = p->max_len < ld ? p->max_len : ld;
  const int maxe = g->max_edges > 0 ? g->max_edges : 1;
  // register-resident walk for graphs of at most 64 nodes (placeholders need bit 15 of a token free);
  // GTOK_SENT_GENERIC=1 forces the LDS bit-matrix kernel (A/B runs, tests of both paths)
  const char *force = std::getenv("GTOK_SENT_GENERIC");
  const bool reg_path = W == 1 && !(force && force[0] == '1') &&
                        22 + GTOK_SENT_IDX_OFFSET + p->max_num_nodes + p->num_node_types + 256 < kEdgeRef &&
                        g->max_edges <= 32768 &&
                        (!p->remap_zinc || g->max_nodes <= p->max_num_nodes);   // remap folded into constants

  SentArgs a;
  a.g = *g; a.p = *p; a.cap = cap; a.maxn = maxn; a.out = out_ids; a.ld = ld; a.out_len = out_len;
  int off = 0;
  if (reg_path) {
    a.l.adj = off; off += 64 * 8;
    a.l.vis = a.l.rng = a.l.vidx = a.l.order = a.l.nat = 0;
    // no store of the walk is bounds-checked: room for the longest possible trail (or max_len) + one iteration
    const int64_t bound = p->labeled ? 2 + 7 * (int64_t)maxn + 2 * (int64_t)maxe : 2 + 5 * (int64_t)maxn + (int64_t)maxe;
    const int tokcap = (int)(bound < p->max_len ? bound : p->max_len) + kSentSlack;
    a.l.tok = off; off += align_up(tokcap * 2, 8);
    a.l.rp = a.l.eat = off;
    a.l.col = off; off += align_up(maxe * 2, 8);          // staged neighbour ids
    if (p->labeled) {
      a.l.rp = off; off += align_up(maxn * maxn, 8);      // edge-type table et[a][b]
      a.l.eat = off; off += align_up(maxe, 8);
    }
  } else {
    a.l.adj = off; off += maxn * W * 8;
    a.l.vis = off; off += W * 8;
    a.l.rng = off; off += 256 * 4;
    a.l.vidx = off; off += align_up(maxn * 2, 8);
    a.l.order = off; off += align_up(maxn * 2, 8);
    a.l.tok = off; off += align_up((cap + 4) * 2, 8);
    a.l.rp = a.l.col = a.l.eat = a.l.nat = off;
    if (p->labeled) {
      a.l.rp = off; off += align_up((maxn + 1) * 4, 8);
      a.l.col = off; off += align_up(maxe * 2, 8);
      a.l.eat = off; off += align_up(maxe, 8);
      a.l.nat = off; off += align_up(maxn, 8);
    }
  }
  a.l.stride = align_up(off, 16);
  if (a.l.stride > 160 * 1024) return GTOK_E_TOO_LARGE;
  int wpb = 4;
  while (wpb > 1 && wpb * a.l.stride > 64 * 1024) wpb >>= 1;
  const size_t lds = (size_t)wpb * a.l.stride;

  void (*kern)(const SentArgs) = nullptr;
#define PICK(w)                                                                   \
  kern = p->labeled ? (void (*)(const SentArgs))sent_kernel<w, true>             \
                    : (void (*)(const SentArgs))sent_kernel<w, false>
  if (reg_path) {
    // no truncation test in the walk when max_len can hold the longest possible trail of this batch
    const int64_t bound = p->labeled ? 2 + 7 * (int64_t)maxn + 2 * (int64_t)maxe : 2 + 5 * (int64_t)maxn + (int64_t)maxe;
    const bool nolim = bound <= p->max_len;
    typedef void (*K)(const SentArgs);
    kern = p->labeled ? (nolim ? (K)sent_reg_kernel<true, true> : (K)sent_reg_kernel<true, false>)
                      : (nolim ? (K)sent_reg_kernel<false, true> : (K)sent_reg_kernel<false, false>);
  } else {
    switch (W) {
      case 1: PICK(1); break;
      case 2: PICK(2); break;
      case 4: PICK(4); break;
      default: PICK(8); break;
    }
  }
#undef PICK
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GTOK_E_LAUNCH;
  }
  int dev = 0, ncu = 256, occ = 1;
  if (hipGetDevice(&dev) != hipSuccess) return GTOK_E_NO_DEVICE;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern),
                                                   wpb * 64, lds) != hipSuccess || occ < 1)
    occ = 1;
  occ = gtok::resident_blocks(occ);
  a.units = (g->num_graphs + wpb - 1) / wpb;
  int nb = ncu * occ;
  if (nb > a.units) nb = a.units;
  a.upb = (a.units + nb - 1) / nb;
  nb = (a.units + a.upb - 1) / a.upb;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(wpb * 64), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}
