// gtok_sent_reg.hpp — SENT walk for graphs with at most 64 nodes (every ZINC molecule), walk state in
// REGISTERS: lane v owns adjacency row v (one 64-bit word of uncovered-edge bits), its visit index and
// its node type; lane k owns the k-th visited node and Philox block k.  Per trail step the wave does
//   v_readlane (row of the current node) -> s_bcnt1 -> Philox word by v_readlane -> s_mul_hi ->
//   v_mbcnt + ballot + s_ff1 (k-th uncovered edge) -> two predicated bit clears
// with no LDS round trip on the dependency chain: tokens go to LDS as fire-and-forget writes, and the
// edge-type tokens of labelled graphs are written as (a,b) placeholders that the row writer resolves for
// all lanes in parallel at the end.  Same spec, same token stream as sent_kernel<1,*>.
#pragma once
#include "gtok_common.hpp"
#include "gtok.h"

namespace gtok {

struct SentLds {  // byte offsets inside a wave's LDS slice, computed on the host
  int adj, vis, vidx, order, rng, tok, rp, col, eat, nat, stride;
};

struct SentArgs {
  gtok_csr g;
  gtok_sent_params p;
  SentLds l;
  int cap;        // min(max_len, ld): trail tokens stored
  int maxn;       // adjacency rows per slice
  int32_t *out;
  int ld;
  int32_t *out_len;
  int epochs;     // K >= 1 (gtok_sent_params.epoch_count): the launch walks G x K (epoch, graph) pairs, epoch-major
  int units;      // ceil(G * K / waves_per_block)
  int upb;        // units per block
  int *queue;     // sent_lds_kernel: ticket counter block (gtok_common.hpp: Tickets)
};

constexpr int kEdgeRef = 0x8000;  // tok entry = kEdgeRef | a << 6 | b : "edge type of (a,b)", resolved by the writer

__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
// number of set bits of a wave-uniform mask below this lane
__device__ __forceinline__ int mbcnt64(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// k-th (0-based) set bit of a wave-uniform word: the lanes whose below-count equals k form the run that
// ends at that bit, so one compare + scalar AND + s_ff1 finds it
__device__ __forceinline__ int kth_bit_reg(uint64_t word, int k) {
  const uint64_t eq = (uint64_t)__ballot(mbcnt64(word) == k);
  return __builtin_ctzll(eq & word);   // k < popcount(word): never empty
}

// Node sets as one 32-bit word when the graph has at most 32 nodes (nearly every ZINC molecule): a row read is one
// v_readlane instead of two, the below-count one v_mbcnt instead of two, and the scalar set arithmetic 32-bit.
template <typename S> __device__ __forceinline__ S readlane_set(S x, int l);
template <> __device__ __forceinline__ uint64_t readlane_set<uint64_t>(uint64_t x, int l) { return readlane64(x, l); }
template <> __device__ __forceinline__ uint32_t readlane_set<uint32_t>(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ int popc_set(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int popc_set(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int ctz_set(uint64_t x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int ctz_set(uint32_t x) { return __builtin_ctz(x); }
__device__ __forceinline__ int mbcnt_set(uint64_t m) { return mbcnt64(m); }
__device__ __forceinline__ int mbcnt_set(uint32_t m) { return (int)__builtin_amdgcn_mbcnt_lo(m, 0u); }   // (lanes >= 32 count all of m: never equal to a k < popcount)
template <typename S> __device__ __forceinline__ int kth_bit_set(S word, int k) {
  const S eq = (S)__ballot(mbcnt_set(word) == k);
  return ctz_set((S)(eq & word));   // k < popcount(word): never empty
}

// Wave-uniform reads of the read-only graph pointers as SCALAR loads (constant address space): the
// compiler will not prove invariance through the by-value argument struct on its own.
typedef const int32_t __attribute__((address_space(4))) *kptr32;
typedef const int64_t __attribute__((address_space(4))) *kptr64;
__device__ __forceinline__ int sload(const int32_t *p, int64_t i) { return ((kptr32)(uintptr_t)p)[i]; }
__device__ __forceinline__ int64_t sload(const int64_t *p, int64_t i) { return ((kptr64)(uintptr_t)p)[i]; }

// remap_zinc_token restricted to node-type tokens (t = node_off + x >= node_off), branch-free:
// train_agtt.py:209-230 — x < ntypes: atom 8+x if x < 9 else 22+t; otherwise the token sits in the edge
// range: b = x - ntypes, bond 17+b if b < 4 else 22+t.
__device__ __forceinline__ int remap_node_type(int x, int node_off, int ntypes) {
  const int t = node_off + x, b = x - ntypes;
  int r = 22 + t;
  r = (x < ntypes && x < 9) ? 8 + x : r;
  r = (b >= 0 && b < 4) ? 17 + b : r;
  return r;
}
// ... and to edge-type tokens (t = edge_off + at >= edge_off): bond 17+at if at < 4 else 22+t
__device__ __forceinline__ int remap_edge_type(int at, int edge_off) { return at < 4 ? 17 + at : 22 + edge_off + at; }

// unsigned forms (token fields are packed with 32-bit operations; a signed int would be sign-extended first)
__device__ __forceinline__ uint32_t remap_node_type_u(uint32_t x, uint32_t node_off, uint32_t ntypes) {
  // x + offset of its range: atoms [0, min(ntypes, 9)) -> 8 + x, the next four ids (they read as bonds) -> 17 + (x - ntypes),
  // everything else 22 + token.  One unsigned compare per range (x - ntypes wraps for x < ntypes), then one add.
  uint32_t off = 22u + node_off;
  off = (x - ntypes) < 4u ? 17u - ntypes : off;
  off = x < (ntypes < 9u ? ntypes : 9u) ? 8u : off;
  return x + off;
}
__device__ __forceinline__ uint32_t remap_edge_type_u(uint32_t at, uint32_t edge_off) { return at < 4u ? 17u + at : 22u + edge_off + at; }

// How much a single walk iteration can append past `lim`: edge + position + type + LADJ + 64 x 2 + RADJ,
// plus RESET/position and EOS, plus the 63 junk slots an unpredicated store touches.  The token buffer is sized min(lim, bound) + kSentSlack, so no store in the
// walk needs a bounds check.
constexpr int kSentSlack = 224;   // + 64 slots for the unpredicated stores

// LAB: labelled graphs.  NOLIM: the host proved max_len >= the longest possible trail of this batch, so
// the walk needs no per-iteration truncation test.
//
// The ZINC remap (trainer/train_agtt.py:171-244) is folded into the emission constants: SOS/RESET/LADJ/
// RADJ/EOS and the position base are picked once per launch, node-type tokens are remapped once per node
// at load time, edge-type tokens once per placeholder in the writer — no per-token remap pass.  (The host
// only takes this kernel with remap when max_nodes <= max_num_nodes, where folding is exact.)
template <bool LAB, bool NOLIM>
__global__ void __launch_bounds__(256) sent_reg_kernel(const SentArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = wave_id(), wpb = (int)(blockDim.x >> 6);
  unsigned char *base = smem + (size_t)wave * a.l.stride;
  uint64_t *adjT = reinterpret_cast<uint64_t *>(base + a.l.adj);   // [64] transposed bits (symmetric closure)
  uint16_t *tok = reinterpret_cast<uint16_t *>(base + a.l.tok);
  uint16_t *colL = reinterpret_cast<uint16_t *>(base + a.l.col);   // staged neighbour ids   (labelled)
  uint8_t *eatL = base + a.l.eat;                                  // staged edge types      (labelled)
  uint8_t *et = base + a.l.rp;                                     // [maxn][maxn] edge type of (a,b), O(1) for the writer
  const int S = a.maxn;

  const int lim = a.p.max_len;
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  const int node_off = idx_off + a.p.max_num_nodes;
  const int edge_off = node_off + a.p.num_node_types;
  const uint32_t k0 = (uint32_t)a.p.seed, k1 = (uint32_t)(a.p.seed >> 32), epoch0 = (uint32_t)a.p.epoch;
  const bool remap = a.p.remap_zinc != 0;
  const bool u16 = (a.p.flags & GTOK_SENT_U16) != 0;               // rows of 16-bit ids (include/gtok.h)
  const int pos_base = remap ? 22 : idx_off;                       // 22 + (t - idx_off)
  const int T_RESET = remap ? 2 : GTOK_SENT_RESET, T_LADJ = remap ? 2 : GTOK_SENT_LADJ;
  const int T_RADJ = remap ? 2 : GTOK_SENT_RADJ, T_EOS = remap ? 1 : GTOK_SENT_EOS;
  constexpr int per = LAB ? 2 : 1;
  const bool is0 = lane == 0, is1 = lane == 1;

  // Software pipeline over the wave's graphs: the scalar pointers of graph i+2 and the per-lane data of
  // graph i+1 (row bounds, node type, first 128 CSR entries) are requested before graph i is walked, so
  // the ~4 dependent memory round trips at the head of every graph overlap with the previous walk.
  struct Ptrs { int n0, n1; int64_t e0, e1; };   // raw values: differences are taken at the use, not at the load
  struct Data { int rs, re, x, c0, c1, a0, a1; };
  auto load_ptrs = [&](int g) -> Ptrs {
    Ptrs p;
    p.n0 = sload(a.g.node_ptr, g); p.n1 = sload(a.g.node_ptr, g + 1);
    p.e0 = sload(a.g.edge_ptr, g); p.e1 = sload(a.g.edge_ptr, g + 1);
    return p;
  };
  auto load_data = [&](int g, const Ptrs &p) -> Data {
    Data q = {0, 0, 0, 0, 0, 0, 0};
    const int32_t *__restrict__ rpg = a.g.rowptr + p.n0 + g;
    const int32_t *__restrict__ colg = a.g.col + p.e0;
    const int pe = (int)(p.e1 - p.e0);
    if (lane < min(p.n1 - p.n0, 64)) {
      q.rs = rpg[lane]; q.re = rpg[lane + 1];
      if (LAB) q.x = a.g.nattr[p.n0 + lane];
    }
    if (lane < pe) { q.c0 = colg[lane]; if (LAB) q.a0 = a.g.eattr[p.e0 + lane]; }
    if (lane + 64 < pe) { q.c1 = colg[lane + 64]; if (LAB) q.a1 = a.g.eattr[p.e0 + lane + 64]; }
    return q;
  };

  const int vb = virtual_block();
  const int u0 = vb * a.upb, u1 = min(a.units, u0 + a.upb);
  // K epochs in one launch: the wave's index runs over (epoch, graph) pairs, epoch-major - pair gv is graph gv mod G in
  // epoch gv / G and row gv of the [K, G, ld] slab.  A wave's pairs are wpb apart: the graph index advances by wpb and
  // wraps (one division per wave, at its first pair).
  const int G = a.g.num_graphs, GV = G * a.epochs;
  auto wrap = [&](int v) -> int { while (v >= G) v -= G; return v; };   // (one subtraction unless the batch has fewer graphs than the workgroup waves)
  int gv = u0 * wpb + wave;
  if (u0 >= u1 || gv >= GV) return;
  int g = gv;
  uint32_t epoch = epoch0;
  if (a.epochs > 1) { const int e0 = gv / G; g = gv - e0 * G; epoch += (uint32_t)e0; }
  Ptrs pc = load_ptrs(g);
  Data dc = load_data(g, pc);
  bool has_next = (u0 + 1 < u1) && (gv + wpb < GV);
  Ptrs pn = pc;
  if (has_next) pn = load_ptrs(wrap(g + wpb));
  for (int unit = u0;; ++unit) {
    Data dn = dc;
    if (has_next) dn = load_data(wrap(g + wpb), pn);
    const bool has_next2 = (unit + 2 < u1) && (gv + 2 * wpb < GV);
    Ptrs pnn = pn;
    if (has_next2) pnn = load_ptrs(wrap(wrap(g + wpb) + wpb));

    const int nfull = pc.n1 - pc.n0;
    const int n = min(nfull, 64);
    const int64_t e0 = pc.e0;
    const int e = min((int)(pc.e1 - pc.e0), a.g.max_edges);
    const int32_t *__restrict__ colg = a.g.col + e0;
    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);

    // ---- this lane's row bounds and node-type TOKEN; stage the entries in LDS (coalesced)
    const int rs = dc.rs, re = dc.re;
    int nat = 0;
    if (LAB && lane < n) nat = remap ? remap_node_type(dc.x, node_off, a.p.num_node_types) : node_off + dc.x;
    adjT[lane] = 0;
    if (lane < e) { colL[lane] = (uint16_t)dc.c0; if (LAB) eatL[lane] = (uint8_t)dc.a0; }
    if (lane + 64 < e) { colL[lane + 64] = (uint16_t)dc.c1; if (LAB) eatL[lane + 64] = (uint8_t)dc.a1; }
    for (int i = lane + 128; i < e; i += kWave) {   // long edge lists: the rest, not prefetched
      colL[i] = (uint16_t)colg[i];
      if (LAB) eatL[i] = a.g.eattr[e0 + i];
    }
    // decision-major Philox: lane j holds the word of decision d0 + j (block (d0+j)>>2, word (d0+j)&3).
    // A graph of n <= 64 nodes takes at most 2n <= 128 decisions: two registers cover every walk.
    auto draws = [&](int d0) -> uint32_t {
      uint32_t o[4];
      philox4x32_10((uint32_t)((d0 + lane) >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
      const int w = lane & 3;
      uint32_t x = o[3];
      x = w == 2 ? o[2] : x;
      x = w == 1 ? o[1] : x;
      x = w == 0 ? o[0] : x;
      return x;
    };
#ifdef GTOK_ABLATE_REG_PHILOX
    uint32_t R = (uint32_t)lane * 2654435761u + gid_lo;
#else
    uint32_t R = draws(0);
#endif
    wave_sync();
    // ---- undirected=True: own row in a register, transposed bits through LDS atomics.  The adjacency is
    // never modified afterwards: for a visited node c the uncovered edges are exactly adj[c] & ~vis.
    // Edge types: et[a][b] = type of the first listed entry a->b, else of the first b->a — reverse cells
    // first, forward cells on top, each lane walking its row backwards so the earliest entry wins.
    int pos = 0;
    auto walk_body = [&](auto set_tag) __attribute__((always_inline)) {
    using set_t = decltype(set_tag);
    constexpr int kSetBits = (int)sizeof(set_t) * 8;
    set_t adj = 0;
    if (a.g.flags & GTOK_CSR_SIMPLE_SYMMETRIC) {
      // host-verified simple undirected graphs listed in both directions: a node's row IS its neighbour set and the
      // type of (a,b) is the type of the listed entry a->b - one pass over the own row, no transposed bits, no atomics
      for (int k0 = rs; k0 < re; k0 += 4) {
        int v[4], at[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = min(k0 + j, re - 1);
          v[j] = colL[k];
          if (LAB) at[j] = eatL[k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k0 + j < re && (unsigned)v[j] < (unsigned)n) {
            adj |= (set_t)1 << v[j];
            if (LAB) et[lane * S + v[j]] = (uint8_t)at[j];
          }
        }
      }
    } else {
    for (int k0 = rs + ((re - rs - 1) & ~3); k0 >= rs; k0 -= 4) {   // chunks of 4, last chunk first
      int v[4], at[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                  // all reads first: one round trip
        const int k = min(k0 + j, re - 1);
        v[j] = colL[k];
        if (LAB) at[j] = eatL[k];
      }
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        if (k0 + j < re && (unsigned)v[j] < (unsigned)n) {
          adj |= (set_t)1 << v[j];
          atomicOr(reinterpret_cast<unsigned long long *>(&adjT[v[j]]), 1ull << lane);
          if (LAB) et[v[j] * S + lane] = (uint8_t)at[j];            // reverse cell
        }
      }
    }
    wave_sync();
    if (LAB) {
      for (int k0 = rs + ((re - rs - 1) & ~3); k0 >= rs; k0 -= 4) {
        int v[4], at[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = min(k0 + j, re - 1);
          v[j] = colL[k];
          at[j] = eatL[k];
        }
#pragma unroll
        for (int j = 3; j >= 0; --j)
          if (k0 + j < re && (unsigned)v[j] < (unsigned)n) et[lane * S + v[j]] = (uint8_t)at[j];   // forward cell on top
      }
    }
    adj |= (set_t)adjT[lane];
    }

    // ---- walk.  The kernel is bound by SCALAR issue (measured: ~1000 SALU vs ~900 VALU per molecule, scalar
    // pipe ~70 % busy), so only what steers control flow lives in SGPRs (vis, cur, its bit, the decision
    // counter); the write cursor, the next position token, the visit counter and the edge-ref prefix are
    // kept as lane-uniform VGPR values (seeded from an opaque zero so the compiler leaves them on the VALU).
    // Token stores are unpredicated: lane j writes slot pos+j, the lanes past the step's 1..3 tokens drop
    // junk into slots that later steps overwrite or that lie beyond the final length.
    // consecutive all-lane stores overlap ACROSS lanes (slot pos+j of one store is slot pos'+j' of the next)
    // although never within a lane, so the compiler must not reorder them: a zero-instruction barrier
#define GTOK_TOK_ORDER() asm volatile("" ::: "memory")
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    set_t vis = 0, bcur = 0;
    int d = 0, cur = 0, ord = 0;
    uint16_t *tokp = tok + 1 + lane;          // slot of this lane at the running position (pos = 1 after SOS)
    int vptok = vz + pos_base;                // position token of the next new node
    int vnv = vz;                             // nodes visited so far
    int vcur6 = vz;                           // kEdgeRef | cur << 6

    auto below = [&](uint32_t nchoices) -> uint32_t {
      const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)R, d & 63);
      ++d;
      if (d == 64) {   // rare: keep the second Philox block out of the common path (opaque to hoisting)
        int d0 = d;
        asm volatile("" : "+s"(d0));
        R = draws(d0);
      }
      return __umulhi(x, nchoices);
    };
    // neighbourhood bracket of the node just visited: LADJ [edge ref] position ... RADJ   (uncommon path)
    auto bracket = [&](int v, set_t A) {
      const int nv = popc_set(vis);
      const int pos = (int)(__builtin_amdgcn_readfirstlane((int)(uintptr_t)tokp) - (int)(uintptr_t)tok) >> 1;
      const bool member = (lane < nv) && ((A >> ord) & (set_t)1);     // lane = visit index: ascending order for free
      const uint64_t M = __ballot(member);
      const int cnt = __popcll(M);
      if (is0) { tok[pos] = (uint16_t)T_LADJ; tok[pos + 1 + per * cnt] = (uint16_t)T_RADJ; }
      if (member) {
        const int q = pos + 1 + per * mbcnt64(M);
        if (LAB) { tok[q] = (uint16_t)(kEdgeRef | (v << 6) | ord); tok[q + 1] = (uint16_t)(pos_base + lane); }
        else tok[q] = (uint16_t)(pos_base + lane);
      }
      GTOK_TOK_ORDER();
      tokp += 2 + per * cnt;
    };
    // first visit of v as a start / restart node (no incoming trail edge): position [type]
    auto start_at = [&](int v) {
      ord = (lane == vnv) ? v : ord;
      const set_t bv = (set_t)1 << v;
      vis |= bv;
      int t = LAB ? __builtin_amdgcn_readlane(nat, v) : vptok;
      t = is0 ? vptok : t;
      *tokp = (uint16_t)t;
      GTOK_TOK_ORDER();
      tokp += per;
      vptok += 1; vnv += 1;
      const set_t A = readlane_set<set_t>(adj, v) & vis;   // only a self loop can be in there
      if (A) bracket(v, A);
      cur = v; bcur = bv; vcur6 = ((vz + v) << 6) | kEdgeRef;
    };

    if (is0) tok[0] = GTOK_SENT_SOS;   // SOS -> <bos>: 0 either way
#ifdef GTOK_ABLATE_REG_WALK   // (profiling builds: no walk - what is left is the per-molecule fixed cost)
    if (n > 1000) {
#else
    if (n > 0) {
#endif
      start_at((int)below((uint32_t)n));
      for (;;) {
        // extend the trail while the current node has an uncovered edge (it always leads to an unvisited node)
        set_t row = readlane_set<set_t>(adj, cur) & ~vis;
        while (row) {
          if (!NOLIM) {
            const int pos = (int)(__builtin_amdgcn_readfirstlane((int)(uintptr_t)tokp) - (int)(uintptr_t)tok) >> 1;
            if (pos >= lim) break;
          }
          const int nxt = kth_bit_set<set_t>(row, (int)below((uint32_t)popc_set(row)));
          ord = (lane == vnv) ? nxt : ord;
          const set_t bn = (set_t)1 << nxt;
          vis |= bn;
          if (LAB) {   // [edge ref] position type in ONE store from lanes 0..2
            int t = __builtin_amdgcn_readlane(nat, nxt);
            t = is1 ? vptok : t;
            t = is0 ? (vcur6 | nxt) : t;
            *tokp = (uint16_t)t;
            GTOK_TOK_ORDER();
            tokp += 3;
          } else {
            *tokp = (uint16_t)vptok;
            GTOK_TOK_ORDER();
            tokp += 1;
          }
          vptok += 1; vnv += 1;
          // uncovered edges back to visited nodes (nxt included: self loop), minus the trail edge just taken
          const set_t adjn = readlane_set<set_t>(adj, nxt);
          const set_t A = adjn & vis & ~bcur;
          if (A) bracket(nxt, A);
          cur = nxt; bcur = bn; vcur6 = ((vz + nxt) << 6) | kEdgeRef;
          row = adjn & ~vis;
        }
        if (!NOLIM) {
          const int pos = (int)(__builtin_amdgcn_readfirstlane((int)(uintptr_t)tokp) - (int)(uintptr_t)tok) >> 1;
          if (pos >= lim) break;
        }
        // dead end: visited nodes that still own uncovered edges
        const set_t live = (set_t)__ballot((adj & ~vis) != 0) & vis;
        const int nv = popc_set(vis);
        if (live) {
          const int c = kth_bit_set<set_t>(live, (int)below((uint32_t)popc_set(live)));
          const int k = __builtin_ctzll((uint64_t)__ballot(lane < nv && ord == c));   // its visit index
          *tokp = (uint16_t)(is0 ? T_RESET : pos_base + k);
          GTOK_TOK_ORDER();
          tokp += 2;
          cur = c; bcur = (set_t)1 << c; vcur6 = ((vz + c) << 6) | kEdgeRef;
          continue;
        }
        if (nv < n) {  // another component or an isolated node
          const set_t un = ~vis & (n >= kSetBits ? (set_t)~(set_t)0 : (set_t)(((set_t)1 << n) - (set_t)1));
          const int c = kth_bit_set<set_t>(un, (int)below((uint32_t)(n - nv)));
          *tokp = (uint16_t)T_RESET;
          GTOK_TOK_ORDER();
          tokp += 1;
          start_at(c);
          continue;
        }
        break;
      }
    }
    *tokp = (uint16_t)T_EOS;
    GTOK_TOK_ORDER();
    pos = ((int)(__builtin_amdgcn_readfirstlane((int)(uintptr_t)tokp) - (int)(uintptr_t)tok) >> 1) + 1;
    };
    if (n <= 32) walk_body((uint32_t)0); else walk_body((uint64_t)0);
#undef GTOK_TOK_ORDER

    // ---- row out: resolve edge-type placeholders, append the query, pad
    const int ltrail = min(pos, lim);
    int len = ltrail;
    if (a.p.query) {  // trainer/train_agtt.py:257-267: [idx_off+N, idx_off+u, idx_off+v] after the trail, not remapped
      if (lane < 3)
        tok[ltrail + lane] = (uint16_t)(idx_off + (lane == 0 ? nfull : a.p.query[2 * (int64_t)g + lane - 1]));
      len = ltrail + 3;
    }
    wave_sync();
    {
      int32_t *__restrict__ orow = a.out + (int64_t)gv * a.ld;
      uint16_t *__restrict__ orow16 = reinterpret_cast<uint16_t *>(a.out) + (int64_t)gv * a.ld;
      const int ld = a.ld, pad = a.p.pad_id, lw = min(len, ld);
      const uintptr_t addr = u16 ? reinterpret_cast<uintptr_t>(orow16) : reinterpret_cast<uintptr_t>(orow);
      if (((ld & 3) == 0) && ((addr & (u16 ? 7u : 15u)) == 0)) {
        for (int i = lane * 4; i < ld; i += kWave * 4) {
          int4 o = make_int4(pad, pad, pad, pad);
          if (i < lw) {
            const uint2 w = *reinterpret_cast<const uint2 *>(tok + i);   // 4 tokens, one ds_read_b64
            int t[4] = {(int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16)};
            if (LAB) {
              bool ph[4];
              int at[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {                              // 4 edge-type reads in flight together
                ph[j] = (t[j] & kEdgeRef) && (i + j < ltrail);
                at[j] = et[ph[j] ? ((t[j] >> 6) & 63) * S + (t[j] & 63) : 0];
              }
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (ph[j]) t[j] = remap ? remap_edge_type(at[j], edge_off) : edge_off + at[j];
            }
            o.x = t[0];
            o.y = i + 1 < lw ? t[1] : pad;
            o.z = i + 2 < lw ? t[2] : pad;
            o.w = i + 3 < lw ? t[3] : pad;
          }
          if (u16) {
            uint2 p2;
            p2.x = ((uint32_t)o.x & 0xFFFFu) | ((uint32_t)o.y << 16); p2.y = ((uint32_t)o.z & 0xFFFFu) | ((uint32_t)o.w << 16);
            *reinterpret_cast<uint2 *>(orow16 + i) = p2;
          } else {
            *reinterpret_cast<int4 *>(orow + i) = o;
          }
        }
      } else {
        for (int i = lane; i < ld; i += kWave) {
          int t = pad;
          if (i < lw) {
            t = tok[i];
            if (LAB && (t & kEdgeRef) && i < ltrail) {
              const int at = et[((t >> 6) & 63) * S + (t & 63)];
              t = remap ? remap_edge_type(at, edge_off) : edge_off + at;
            }
          }
          if (u16) orow16[i] = (uint16_t)t; else orow[i] = t;
        }
      }
    }
    if (is0) a.out_len[gv] = len;
    wave_sync();
    if (!has_next) break;
    gv += wpb;
    for (g += wpb; g >= G; g -= G) ++epoch;      // into the next epoch's slice
    pc = pn; dc = dn; pn = pnn;
    has_next = has_next2;
  }
}

}  // namespace gtok
