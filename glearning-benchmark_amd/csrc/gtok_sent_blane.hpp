// gtok_sent_blane.hpp — SENT walk, LANE per graph, for UNLABELLED graphs of up to 256 nodes of any shape (self loops,
// duplicate or one-directional edge lists), over a resident ADJACENCY BIT-MATRIX mirror of the batch.
//
// The wave-per-graph kernels spend ~120 instructions per trail step on ONE graph and are bound by the CU's single scalar
// pipe; the lane-per-graph kernel of gtok_sent_lane.hpp advances 64 walks per instruction but keeps each graph's CSR in
// LDS, which stops at molecule size.  Here a graph's adjacency lives in HBM as an immutable bit matrix (rows of W = 1, 2
// or 4 64-bit words: the symmetric closure of the edge list, built ONCE per resident batch by gtok_csr_adjbits, like the
// CSR itself), and a lane fetches the one row it needs per step (8 W bytes, the only HBM read of the walk).  Walk state is
// all registers: the visited set (W words), the row of the current node, and P x W words of bit-sliced counters (plane p
// = bit p of every node's number of unvisited neighbours; their initial value - the degrees - comes with the mirror), from
// which `live` (visited nodes that still own an uncovered edge) is formed only when a lane is at a dead end.  The one
// per-node table, node -> visit index (position tokens of bracket members and restarts), sits in LDS laid out
// [dword][lane]: lane-private, bank = lane, conflict-free.  A bracket is listed in ascending visit index by passing its
// members (a set in node space) through a set in visit-index space, both streamed word by word through LDS so that every
// lane iterates over its own members only.  Same spec and token stream as every
// other SENT kernel, bit-exact against oracle/gtok_oracle.c:oracle_sent.
//
// Lanes of a wave need not hold neighbouring graphs: `lane_order` (optional, part of the mirror) lists the graphs in the
// order they are dealt to lanes, so that a wave's 64 walks have similar lengths (a unit lasts as long as its longest walk).
#pragma once
#include "gtok_sent_lane.hpp"

namespace gtok {

struct SentBLaneArgs {
  gtok_csr g;
  gtok_sent_params p;
  int32_t *out;
  int ld;
  int32_t *out_len;
  int units;   // ceil(G / 64): units of ONE epoch
  int epochs;  // K >= 1 (gtok_sent_params.epoch_count): the launch walks units x K (unit, epoch) pairs, unit-major
  int prio;    // long units at a higher issue priority than the short ones they share a SIMD with
  int pad_nt;  // padding leaves with non-temporal stores (slabs larger than the memory-side cache: gtok_sent_lane.hpp)
  int epoch_major;   // order of the (unit, epoch) pairs (gtok_sent_lane.hpp)
  int *tickets;      // 2 ints per workgroup in device memory, zero between launches: [2 b] the ticket of the pairs beyond the first
                     // round, [2 b + 1] the waves of workgroup b that are through (the last one re-arms both).  Round 5: the LDS is
                     // full to the last byte (8 x 20 KB at W = 4), the counter moved out; a draw costs ~1 us once per ~400 us unit
};

struct __attribute__((aligned(16))) U64x2 { uint64_t a, b; };
#ifndef GTOK_BLANE_CONV
#define GTOK_BLANE_CONV 4
#endif
constexpr int kBlaneConv = GTOK_BLANE_CONV;   // bracket members taken to visit-index space per loop iteration

// U16: the GTOK_SENT_U16 slab - rows of 16-bit ids, the token windows stored as they stand
template <int W, int P, bool U16>
__global__ void __launch_bounds__(W == 4 ? 512 : 1024) sent_blane_kernel(const SentBLaneArgs a) {
  using out_t = typename std::conditional<U16, uint16_t, int32_t>::type;
  constexpr int EV = U16 ? 8 : 4;            // ids per 16-byte store
  out_t *const out_base = reinterpret_cast<out_t *>(a.out);
  // LDS, all of it lane-private and laid out [dword][lane] (bank = lane): 16 W dwords node -> visit index (u8 each),
  // 2 W dwords bracket members in visit-index space (zero between brackets), 2 W dwords the bracket's members in node space
  // (staged there so that a lane can fetch ITS next non-empty sub-word with one indexed LDS read: see the bracket loops)
  // one workgroup per CU (8 waves at W = 4, else 16), each wave with its own 20 W x 256 bytes of the workgroup's LDS;
  // waves never cooperate
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int lane = lane_id(), wave = wave_id();
  unsigned char *smem = smem_all + (size_t)wave * (20 * W * 256);
  // (the visit-index table is [node][lane] BYTES: one shift-add per address - the [dword][lane] form of rounds 2-4 was free of bank
  // conflicts but cost four instructions per look-up, and this kernel's time is its instruction count, not its LDS passes)
  uint8_t *vx = smem + lane;
  auto vx_at = [&](int u) __attribute__((always_inline)) -> uint8_t & { return vx[u << 6]; };
  uint32_t *lw = reinterpret_cast<uint32_t *>(smem) + lane;
  constexpr int TW0 = 16 * W * 64;   // dword offset of the visit-index set
  constexpr int MW0 = TW0 + 2 * W * 64;   // ... of the staged member set (node space)

  const int lim = a.p.max_len, ld = a.ld, cap = min(lim, ld);
  const int idx_off = GTOK_SENT_IDX_OFFSET;
  uint32_t k0, k1;   // Philox key in vector registers (gtok_sent_lane.hpp)
  asm volatile("v_mov_b32 %0, %1" : "=v"(k0) : "s"((uint32_t)a.p.seed));
  asm volatile("v_mov_b32 %0, %1" : "=v"(k1) : "s"((uint32_t)(a.p.seed >> 32)));
  const uint32_t epoch0 = (uint32_t)a.p.epoch;
  const uint64_t T_RESET = GTOK_SENT_RESET, T_LADJ = GTOK_SENT_LADJ, T_RADJ = GTOK_SENT_RADJ, T_EOS = GTOK_SENT_EOS;
  const int G = a.g.num_graphs, pad = a.p.pad_id;
  const bool no_pad = (a.p.flags & GTOK_SENT_NO_PAD) != 0, pad_nt = a.pad_nt != 0;

  // Units are stored by descending expected length (lane_order), and the waves w, w + 4, w + 8, ... of a workgroup share a
  // SIMD (observed on gfx950, profiles/tools/probes/wave_simd_probe.hip; speed only): per round of gridDim x waves units,
  // wave w takes rank (w >> 2) * qsize + j of that order, j = (w & 3) * gridDim + block, odd groups backwards - every SIMD
  // holds long units together with short ones, the long ones at the higher issue priority.
  const int nwaves = (int)(blockDim.x >> 6), grp = wave >> 2, qsize = (int)gridDim.x * 4, jq = (wave & 3) * (int)gridDim.x + (int)blockIdx.x;
  __builtin_amdgcn_s_setprio(0);
  if (!a.prio) {
  } else if (nwaves > 8) {
    if (grp == 0) __builtin_amdgcn_s_setprio(3);
    else if (grp == 1) __builtin_amdgcn_s_setprio(2);
    else if (grp == 2) __builtin_amdgcn_s_setprio(1);
  } else if (nwaves > 4 && grp == 0) {
    __builtin_amdgcn_s_setprio(1);
  }
  // K epochs per launch: the launch walks (unit, epoch) pairs - unit-major, or epoch-major when one epoch fills half of the
  // resident waves or more (gtok_sent_lane.hpp).  The first round of resident waves is dealt statically as described above;
  // pairs beyond it are drawn from a ticket counter in the workgroup's LDS (workgroup b owns pairs nslots + t * gridDim + b,
  // longest first), at priority 0: a wave that has finished takes the longest pair left instead of leaving its SIMD one wave
  // short (125 k ER graphs x 2 epochs: 0.44 ms per epoch with the static deal already, against 0.51 for one epoch per launch).
  const int K = a.epochs, vunits = a.units * K, nslots = (int)gridDim.x * nwaves;
  int *const wg_ticket = a.tickets + 2 * (int)blockIdx.x;
  for (int vu = grp * qsize + ((grp & 1) ? qsize - 1 - jq : jq); vu < vunits;) {
    int unit = vu, ep = 0;
    if (K > 1) {
      if (a.epoch_major) { ep = vu / a.units; unit = vu - ep * a.units; }
      else { unit = vu / K; ep = vu - unit * K; }
    }
    const uint32_t epoch = epoch0 + (uint32_t)ep;
    const int64_t row0 = (int64_t)ep * G;          // first row of the epoch's [G, ld] slice
    const int slot = unit * 64 + lane;
    const bool valid = slot < G;
    const int g = valid ? (a.g.lane_order ? a.g.lane_order[slot] : slot) : 0;
    int nb0 = 0, nfull = 0;
    if (valid) { nb0 = a.g.node_ptr[g]; nfull = a.g.node_ptr[g + 1] - nb0; }
    const int n = min(nfull, 64 * W);
    const uint64_t *__restrict__ rows = a.g.adj_rows + (size_t)nb0 * W;
    auto load_row = [&](int v, uint64_t (&r)[W]) __attribute__((always_inline)) {
      const uint64_t *q = rows + (size_t)v * W;
      if (W == 1) { r[0] = q[0]; }
      else {
#pragma unroll
        for (int w = 0; w < W; w += 2) { const U64x2 x = *reinterpret_cast<const U64x2 *>(q + w); r[w] = x.a; r[w + 1] = x.b; }
      }
    };
    // counters: plane p, word w
    uint64_t c[P][W];
    {
      const uint64_t *pl = a.g.adj_planes + (size_t)g * 8 * W;
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int w = 0; w < W; ++w) c[p][w] = valid ? pl[p * W + w] : 0ull;
    }
    // planes in use in this unit (the same in every lane): counts only go down, so a plane that starts all zero stays so
    int peff = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      uint64_t any = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) any |= c[p][w];
      if (__ballot(any != 0)) peff = p + 1;
    }
    uint64_t validm[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { const int r = n - 64 * w; validm[w] = r >= 64 ? ~0ull : (r > 0 ? ((1ull << r) - 1ull) : 0ull); }

    const uint64_t gid = (uint64_t)(a.p.graph_base + g);
    const uint32_t gid_lo = (uint32_t)gid, gid_hi = (uint32_t)(gid >> 32);
    out_t *__restrict__ orow = out_base + (row0 + g) * ld;
    uint64_t vis[W], rowc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { vis[w] = 0; rowc[w] = 0; }
#pragma unroll
    for (int k = 0; k < 2 * W; ++k) lw[TW0 + k * 64] = 0;
    uint64_t wlo = 0;
    int nvis = 0, pos = 0, fl = 0, d = 0, cur = 0;

    // ---- token window as in gtok_sent_lane.hpp, flushed four tokens (one 16-byte store) at a time: this kernel is bound
    // by the instructions of its bracket loops, not by its HBM writes
#ifndef GTOK_BLANE_SECTOR_GROUPS
#define GTOK_BLANE_SECTOR_GROUPS 1
#endif
    constexpr int SG = GTOK_BLANE_SECTOR_GROUPS;
    uint64_t pg[SG > 1 ? SG - 1 : 1];
#pragma unroll
    for (int j = 0; j < SG - 1; ++j) pg[j] = 0;
    auto tok_of = [](uint64_t w, int i) __attribute__((always_inline)) -> int { return (int)((w >> (i << 4)) & 0xFFFFu); };
    auto put4 = [&](int at, uint64_t w) __attribute__((always_inline)) { store_win(orow + at, w); };
    auto group_of = [&](int j, uint64_t w) __attribute__((always_inline)) -> uint64_t {
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
        if (fl + 4 <= cap && (ld & 3) == 0) {
#pragma unroll
          for (int k = 0; k < SG - 1; ++k) put4(sb + 4 * k, pg[k]);
          put4(fl, w);
        } else {
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
    uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
    auto below = [&](uint32_t nchoices) __attribute__((always_inline)) -> uint32_t {   // d is the same in every active lane
      const int w = uni(d) & 3;
      if (w == 0) {
        uint32_t o[4];
        philox4x32_10((uint32_t)(d >> 2), epoch, gid_lo, gid_hi, k0, k1, o);
        pw0 = o[0]; pw1 = o[1]; pw2 = o[2]; pw3 = o[3];
      }
      const uint32_t x = w == 0 ? pw0 : (w == 1 ? pw1 : (w == 2 ? pw2 : pw3));
      ++d;
      return __umulhi(x, nchoices);
    };
    // k-th member (ascending node id) of a W-word set, k below its size
    auto kth_of = [&](const uint64_t (&s)[W], int k) __attribute__((always_inline)) -> int {
      uint64_t word = s[W - 1];
      int base = 64 * (W - 1), kk = k;
      bool found = false;
#pragma unroll
      for (int w = 0; w < W - 1; ++w) {
        const int cw = __popcll(s[w]);
        const bool take = !found && kk < cw;
        word = take ? s[w] : word;
        base = take ? 64 * w : base;
        found = found || take;
        kk = found ? kk : kk - cw;
      }
      return base + kth_bit64(word, kk);
    };
    auto count_of = [&](const uint64_t (&s)[W]) __attribute__((always_inline)) -> int {
      int t = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) t += __popcll(s[w]);
      return t;
    };
    // ---- a set of 64 W bits consumed in ascending order OUT OF THE LANE'S LDS (2 W dwords at dword offset `off`, [dword][lane]),
    // four members per loop iteration and NO BRANCH inside an iteration.  A lane holds two 32-bit sub-words of its set in
    // registers (`cur`, then `nxt`); `nz` = the non-empty sub-words beyond those.  An iteration
    //   top     asks LDS for the sub-word after `nxt` (the index differs from lane to lane, the bank does not) - always, used or not;
    //   middle  pops up to four members from cur / nxt with selects (a pop from an empty pair is harmless and masked by the
    //           caller's count), as many as the pair holds: stream_take() says how many;
    //   bottom  shifts the pair along when `cur` is used up - selects again - and only there waits for the read of the top.
    // Every lane moves through ITS non-empty sub-words, so a wave runs as many iterations as its largest set needs, whatever
    // words the members fall in.  (Rounds 2-4 kept the set in W registers and picked sub-words with a tree of bit-field inserts
    // under masks; an LDS stream with a refill branch behind every pop waited for its read inside the branch - the compiler
    // has to finish the register shuffle before the join.  Both cost ~130 vector instructions per four members, and since
    // SOME lane of the wave needed the slow path at nearly every pop, every iteration paid all of them.)
    struct Stream { uint32_t cur, nxt, nz, pre; int base, nbase, pk; };
    auto sub_slot = [&](int off, uint32_t nz, int &k) __attribute__((always_inline)) -> uint32_t * {
      k = max(__ffs((int)nz) - 1, 0);          // (nz == 0: sub-word 0 - a valid address, a value nobody uses)
      return lw + off + k * 64;
    };
    auto stream_open = [&](Stream &st, int off, uint32_t nz) __attribute__((always_inline)) {   // nz != 0
      int k0, k1;
      const uint32_t c = *sub_slot(off, nz, k0);
      nz &= nz - 1;
      const uint32_t n = *sub_slot(off, nz, k1);
      st.cur = c; st.base = k0 << 5;
      st.nxt = nz ? n : 0u; st.nbase = k1 << 5;
      st.nz = nz & (nz - 1);
    };
    auto stream_top = [&](Stream &st, int off) __attribute__((always_inline)) { st.pre = *sub_slot(off, st.nz, st.pk); };
    auto stream_take = [&](const Stream &st, int left) __attribute__((always_inline)) -> int {   // members this iteration can pop
      return min(min(left, 4), __popc(st.cur) + __popc(st.nxt));
    };
    auto stream_pop = [&](Stream &st) __attribute__((always_inline)) -> int {
      const bool fc = st.cur != 0;
      const uint32_t src = fc ? st.cur : st.nxt;
      int low;                                 // lowest set bit; -1 for an empty pair (v_ffbl_b32: defined, unlike __builtin_ctz(0))
      asm("v_ffbl_b32 %0, %1" : "=v"(low) : "v"(src));
      const int u = (fc ? st.base : st.nbase) + low;
      st.cur &= st.cur - 1;                    // (0 stays 0)
      st.nxt = fc ? st.nxt : (src & (src - 1));
      return u;
    };
    auto stream_bottom = [&](Stream &st) __attribute__((always_inline)) {
      const bool adopt = st.cur == 0;
      const uint32_t fresh = st.nz ? st.pre : 0u;
      st.cur = adopt ? st.nxt : st.cur; st.base = adopt ? st.nbase : st.base;
      st.nxt = adopt ? fresh : st.nxt; st.nbase = adopt ? (st.pk << 5) : st.nbase;
      st.nz = adopt ? (st.nz & (st.nz - 1)) : st.nz;
    };

#ifdef GTOK_PHASE_TIMING   // profiling build only: cycles per phase, left in the last 8 columns of the row of the unit's lane 0
    uint64_t pt[5] = {0, 0, 0, 0, 0}, pt_last = __builtin_amdgcn_s_memtime();
    uint32_t pt_steps = 0, pt_it1 = 0, pt_it2 = 0;
    const uint32_t pt_rt0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#define GTOK_PT(i) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); pt[i] += now_ - pt_last; pt_last = now_; }
#else
#define GTOK_PT(i)
#endif
    if (valid) {
      append((uint64_t)GTOK_SENT_SOS, 1);
      if (n > 0) {
        // The loop is software-pipelined: with the row of step k in hand (counters, visited set and bracket members
        // brought up to date), the node of step k + 1 is chosen and its row requested BEFORE step k's tokens are written
        // - the choice does not depend on them - so that the one HBM access of a step travels behind the bracket loops.
        int kind = 0, pick = 0, after = pos;   // `after`: the row's length once the current step's tokens are out
        uint32_t xb = 0;
        bool have = false;                      // false in the first pass only (same in every lane: nothing to write yet)
        for (;;) {
          uint64_t M[W], tokv = 0;
          int tokc = 0, nm = 0;      // nm: members of this step's bracket
          bool anym = false;
#pragma unroll
          for (int w = 0; w < W; ++w) M[w] = 0;
          if (have) {
#ifdef GTOK_PHASE_TIMING
            ++pt_steps;
#endif
            const bool first = kind != 1;
            const uint32_t my = (uint32_t)nvis;
            if (first) vx_at(pick) = (uint8_t)my;
            {
              const uint64_t tpos = (uint64_t)((uint32_t)idx_off + (first ? my : xb));   // xb: visit index of a restart node
              const bool has_a = kind == 1 || (kind == 2 && nvis > 0);   // (the walk's first node is a component start without RESET)
              tokv = has_a ? (T_RESET | (tpos << 16)) : tpos;
              tokc = has_a ? 2 : 1;
            }
            // first visit: the node's neighbours lose an unvisited neighbour; visited neighbours (itself included: self
            // loop) other than the trail's predecessor form its bracket
            uint64_t bw[W];
#pragma unroll
            for (int w = 0; w < W; ++w) bw[w] = first ? rowc[w] : 0ull;
#pragma unroll
            for (int p = 0; p < P; ++p)
              if (p < peff) {
#pragma unroll
                for (int w = 0; w < W; ++w) { const uint64_t t = c[p][w]; c[p][w] = t ^ bw[w]; bw[w] &= ~t; }
              }
#pragma unroll
            for (int w = 0; w < W; ++w) {
              const uint64_t S = first ? rowc[w] : 0ull;
              const uint64_t pickbit = (pick >> 6) == w ? 1ull << (pick & 63) : 0ull;
              const uint64_t predbit = (kind == 0 && (cur >> 6) == w) ? 1ull << (cur & 63) : 0ull;
              vis[w] |= pickbit;
              M[w] = S & vis[w] & ~predbit;
              nm += __popcll(M[w]);
            }
            anym = nm != 0;
            nvis += first;
            if (anym) { tokv |= T_LADJ << (tokc << 4); ++tokc; }
            after = pos + tokc + (anym ? nm + 1 : 0);
            GTOK_PT(1)
          }
          // ---- the next node: rowc = row of the node the trail stands on (all zero before the first step)
          bool more = after < lim;
          int nkind = 0, npick = 0;
          uint32_t nxb = 0;
          uint64_t rn[W];
#pragma unroll
          for (int w = 0; w < W; ++w) rn[w] = 0;
          if (more) {
            uint64_t set[W];
            int cnt = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) { set[w] = rowc[w] & ~vis[w]; cnt += __popcll(set[w]); }
            if (cnt == 0) {   // dead end: visited nodes that still own an uncovered edge, else another component / isolated node
              uint64_t nz[W];
#pragma unroll
              for (int w = 0; w < W; ++w) nz[w] = 0;
#pragma unroll
              for (int p = 0; p < P; ++p)
                if (p < peff) {
#pragma unroll
                  for (int w = 0; w < W; ++w) nz[w] |= c[p][w];
                }
#pragma unroll
              for (int w = 0; w < W; ++w) { set[w] = vis[w] & nz[w]; cnt += __popcll(set[w]); }
              nkind = 1;
              if (cnt == 0) {
                nkind = 2;
#pragma unroll
                for (int w = 0; w < W; ++w) { set[w] = ~vis[w] & validm[w]; cnt += __popcll(set[w]); }
              }
            }
            more = cnt != 0;                                     // else: every node visited, every edge covered
            if (more) {
              npick = kth_of(set, (int)below((uint32_t)cnt));
              load_row(npick, rn);
              nxb = vx_at(npick);
            }
          }
          GTOK_PT(0)
          // ---- the current step's tokens
          if (have) {
            append(tokv, tokc);
            if (anym && pos < lim) {
              // LADJ, members by ascending visit index, RADJ: each member (node space, any order) sets its visit index's
              // bit in the lane's LDS set (two per iteration: two independent LDS reads in flight) ...
              uint32_t nzm = 0;
#pragma unroll
              for (int w = 0; w < W; ++w) {
                const uint32_t lo = (uint32_t)M[w], hi = (uint32_t)(M[w] >> 32);
                lw[MW0 + (2 * w) * 64] = lo; lw[MW0 + (2 * w + 1) * 64] = hi;
                nzm |= (lo ? 1u << (2 * w) : 0u) | (hi ? 2u << (2 * w) : 0u);
              }
              Stream sm;
              stream_open(sm, MW0, nzm);
              uint32_t nzt = 0;
              for (int left = nm; left > 0;) {
                stream_top(sm, MW0);
                const int c4 = stream_take(sm, left);
                int u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int x = stream_pop(sm); u[j] = (j == 0 || j < c4) ? x : u[0]; }   // (a lane with fewer repeats its first)
                uint32_t t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = vx_at(u[j]);
                left -= c4;
                stream_bottom(sm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  __hip_atomic_fetch_or(&lw[TW0 + (t[j] >> 5) * 64], 1u << (t[j] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  nzt |= 1u << (t[j] >> 5);
                }
#ifdef GTOK_PHASE_TIMING
                ++pt_it1;
#endif
              }
              GTOK_PT(2)
              // ... which is then streamed back and listed in ascending order, up to a whole 4-token window per iteration, RADJ
              // behind the last; then the set is zeroed for the next bracket
              Stream stt;
              stream_open(stt, TW0, nzt);
              for (int left = nm; left > 0 && pos < lim;) {
                stream_top(stt, TW0);
                const int c4 = stream_take(stt, left);
                uint32_t tk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t x = (uint32_t)(idx_off + stream_pop(stt)); tk[j] = j < c4 ? x : 0u; }
                left -= c4;
                uint64_t val = ((uint64_t)(tk[2] | (tk[3] << 16)) << 32) | (tk[0] | (tk[1] << 16));
                int cntt = c4;
                const bool last = left == 0;
                if (last && c4 < 4) { val |= T_RADJ << (c4 << 4); ++cntt; }
                append(val, cntt);
                if (last && c4 == 4) append(T_RADJ, 1);
                stream_bottom(stt);
#ifdef GTOK_PHASE_TIMING
                ++pt_it2;
#endif
              }
#pragma unroll
              for (int k = 0; k < 2 * W; ++k) lw[TW0 + k * 64] = 0;
              GTOK_PT(3)
            }
          }
          if (!more) break;
          cur = pick; pick = npick; kind = nkind; xb = nxb;
#pragma unroll
          for (int w = 0; w < W; ++w) rowc[w] = rn[w];
          have = true;
        }
      }
      append(T_EOS, 1);
    }
    // ---- end of the row (as in gtok_sent_lane.hpp): window, query tail, padding to a 16-token boundary
    const int len = min(pos, lim);
    int tot = len;
    int q0 = 0, q1 = 0, q2 = 0;
    if (valid && a.p.query) {
      q0 = idx_off + nfull; q1 = idx_off + a.p.query[2 * (int64_t)g]; q2 = idx_off + a.p.query[2 * (int64_t)g + 1];
      tot = len + 3;
    }
    int padfrom = 0;
    if (valid) {
      a.out_len[row0 + g] = tot;
      const int sb = fl & ~(4 * SG - 1);
      if (!a.p.query && pos <= cap && (ld & 3) == 0) {
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
            if (i < sb) continue;
            v = tok_of(i >= fl ? wlo : group_of((i - sb) >> 2, 0), i & 3);
          } else if (i < tot) {
            v = i == len ? q0 : (i == len + 1 ? q1 : q2);
          }
          orow[i] = (out_t)v;
        }
      }
    }
    // ---- pad tails: four rows per pass, 16 lanes x 16-byte stores on each (rows of a unit need not be neighbours)
    if (!no_pad) {
      auto pad_tails = [&](auto nt_tag) __attribute__((always_inline)) {
        constexpr bool NT = decltype(nt_tag)::value;
        const int q = lane & 15;
        const gtok_v4i pv = pad_vec(out_base, pad);
        for (int it = 0; it < 16; ++it) {
          const int r = it * 4 + (lane >> 4);
          const int lr = __builtin_amdgcn_ds_bpermute(r << 2, padfrom);
          const int gr = __builtin_amdgcn_ds_bpermute(r << 2, valid ? (int)(row0 + g) : -1);
          if (unit * 64 + it * 4 >= G) break;
          if (gr >= 0) {
            out_t *__restrict__ rowp = out_base + (int64_t)gr * ld + lr;
            const int nrem = ld - lr, nvec = nrem / EV;
            _Pragma("clang loop vectorize(disable) unroll(disable)")
            for (int t = q; t < nvec; t += 16) store_pad16<NT>(rowp + EV * t, pv);
            if (q < (nrem & (EV - 1))) rowp[nvec * EV + q] = (out_t)pad;
          }
        }
      };
      if (pad_nt) pad_tails(std::true_type{});
      else pad_tails(std::false_type{});
    }
    __builtin_amdgcn_wave_barrier();
#ifdef GTOK_PHASE_TIMING
    GTOK_PT(4)
    if (!U16 && lane == 0 && ld >= 16) {
      int32_t *row = a.out + (row0 + g) * ld + ld - 8;
      row[0] = (int32_t)pt[0]; row[1] = (int32_t)pt[1]; row[2] = (int32_t)pt[2]; row[3] = (int32_t)pt[3]; row[4] = (int32_t)pt[4];
      row[5] = (int32_t)uni(pt_steps); row[6] = (int32_t)((uni(pt_it1) << 16) | uni(pt_it2));
      row[7] = (int32_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - pt_rt0);
    }
#endif
    // ---- the next pair
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(wg_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    vu = nslots + uni(t) * (int)gridDim.x + (int)blockIdx.x;
    __builtin_amdgcn_s_setprio(0);
  }
  // every wave passes here exactly once; the workgroup's last one leaves both counters zero for the launch that gets them next
  if (lane == 0 && __hip_atomic_fetch_add(wg_ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwaves - 1) {
    __hip_atomic_store(wg_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(wg_ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// the mirror: adjacency bit matrix (symmetric closure of the listed entries, self loops kept) + degree planes.
// Wave per graph: the matrix is assembled in LDS with atomic ORs (lane = row, 16 neighbour loads in flight), written
// out coalesced; node u's degree = popcount of its row, its bit p goes to plane p (one ballot per plane and word).
// ---------------------------------------------------------------------------------------------
struct AdjBitsArgs {
  gtok_csr g;
  int W;
  uint64_t *rows, *planes;
  int32_t *info;   // [0] = largest degree in the symmetric closure (self loop counted): picks the number of counter planes
};

template <int W>
__global__ void __launch_bounds__(256) adj_bits_kernel(const AdjBitsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id(), wave = wave_id(), wpb = (int)(blockDim.x >> 6);
  uint64_t *adj = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * (64 * W * W);
  const int G = a.g.num_graphs;
  for (int g = (int)blockIdx.x * wpb + wave; g < G; g += (int)gridDim.x * wpb) {
    const int nb0 = sload(a.g.node_ptr, g);
    const int n = min(sload(a.g.node_ptr, g + 1) - nb0, 64 * W);
    const int64_t e0 = sload(a.g.edge_ptr, g);
    const int32_t *__restrict__ rpg = a.g.rowptr + nb0 + g;
    const int32_t *__restrict__ colg = a.g.col + e0;
    wave_sync();
    for (int i = lane; i < n * W; i += kWave) adj[i] = 0;
    wave_sync();
    for (int u = lane; u < n; u += kWave) {
      const int rs = rpg[u], re = rpg[u + 1];
      for (int k0e = rs; k0e < re; k0e += 16) {
        int v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = colg[min(k0e + j, re - 1)];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (k0e + j < re && (unsigned)v[j] < (unsigned)n) {
            atomicOr(reinterpret_cast<unsigned long long *>(&adj[u * W + (v[j] >> 6)]), 1ull << (v[j] & 63));
            atomicOr(reinterpret_cast<unsigned long long *>(&adj[v[j] * W + (u >> 6)]), 1ull << (u & 63));
          }
        }
      }
    }
    wave_sync();
    uint64_t *__restrict__ dst = a.rows + (size_t)nb0 * W;
    for (int i = lane; i < n * W; i += kWave) dst[i] = adj[i];
    uint64_t *__restrict__ pl = a.planes + (size_t)g * 8 * W;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int u = 64 * w + lane;
      int deg = 0;
      if (u < n) {
#pragma unroll
        for (int x = 0; x < W; ++x) deg += __popcll(adj[u * W + x]);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint64_t m = __ballot((deg >> p) & 1);
        if (lane == 0) pl[p * W + w] = m;
      }
      int mx = deg;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
      if (lane == 0 && mx > 0) atomicMax(a.info, mx);
    }
  }
}

}  // namespace gtok
