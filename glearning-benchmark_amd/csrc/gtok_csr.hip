// gtok_csr.hip — layout steps on a batch that is already resident in HBM (no reference counterpart: the reference holds
// one PyG `Data` per graph; these are the device-side equivalents of what graph_data_loader/zinc_dataset_autograph.py:51-73
// hands to the tokenizer at trainer/train_agtt.py:246-250, prepared once per split instead of once per item):
//
//   gtok_csr_check       verifies what GTOK_CSR_SIMPLE_SYMMETRIC claims (no self loop, no entry listed twice, (v,u) listed
//                        whenever (u,v) is) and measures max_nodes / max_edges / max_degree - what the Python mirror used to
//                        establish with two torch sorts, and a C caller could not establish at all
//   gtok_csr_lane_sort   the reordered copy sent_lane_kernel walks fastest (graphs by descending nodes + leaves, cut into
//                        units of <= 64 graphs whose staged bytes fit a wave's LDS share: graph_ids / unit_ptr / unit_info,
//                        the permuted CSR arrays and their byte mirror) - counting sort + single-pass scans + a
//                        block-wise pointer chase, twelve small launches, no host round trip
//
// Everything here is integer bookkeeping over arrays of a few MB: the kernels are bound by launch gaps and by the
// latency of dependent loads, not by HBM; their job is to cost well under one epoch of tokenization, once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtok.h"
#include "gtok_common.hpp"

namespace gtok {

constexpr int kInfoViol = 0, kInfoMaxDeg = 1, kInfoMaxN = 2, kInfoMaxE = 3, kInfoUnits = 4, kInfoChunkN = 5, kInfoChunkE = 6;

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void raise_max(int32_t *p, int v) {
  if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}

// ---------------------------------------------------------------------------------------------------------------
// structure check + per-graph sort key.  One wave per 64 consecutive graphs; lanes stride over the group's ROWS (the
// rows of a group are contiguous in rowptr, so the loads coalesce), a row's graph is found by bisection over the group's
// 65 node offsets in LDS.
// ---------------------------------------------------------------------------------------------------------------
struct CsrScanArgs {
  const int32_t *node_ptr; const int64_t *edge_ptr; const int32_t *rowptr; const int32_t *col;
  int G;
  int32_t *info;    // [violations, max degree, max nodes, max entries, ...]: accumulated with atomics
  uint8_t *keys;    // optional: min(255, nodes + rows of length 1) per graph - the expected walk length (a walk restarts once per dead end)
  int cap_r, cap_e; // bytes of LDS per wave for a group's row pointers / entries staged as BYTES (0: no staged path)
};

// A group of 64 small graphs (every row pointer and neighbour id fits a byte, the group's arrays fit the wave's LDS slice) is
// staged with coalesced 16-byte loads - all in flight at once - and examined lane per graph from LDS: the rows of a molecule
// are ~100 contiguous bytes, and the version that went to HBM for every row (bisection for the row's graph, then two
// dependent loads, 25 times in a row per lane) took 153 us for the keys of ZINC-full and 259 us with the entry checks.
// Values that do not fit a byte are counted as violations while staging (the batch is then not what the caller claimed).
template <bool CHECK>
__device__ __forceinline__ void scan_group_staged(const CsrScanArgs &a, const int32_t *np, const int64_t *ep, int g0, int cnt, int n_l, int e_l,
                                                  uint8_t *srp, uint8_t *scol, int &viol, int &maxdeg, int &leaves) {
  const int lane = lane_id();
  const int N0 = np[0], cr = np[cnt] - N0 + cnt;
  const int64_t E0 = ep[0];
  const int ce = (int)(ep[cnt] - E0);
  auto stage = [&](const int32_t *__restrict__ src, int count, uint8_t *dst) {
    uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
    const int nv = count >> 2;
    const I32x4 *v4 = reinterpret_cast<const I32x4 *>(src);
    for (int t = lane; t < nv; t += kWave) {
      const I32x4 v = v4[t];
      viol += (((uint32_t)v.x | (uint32_t)v.y | (uint32_t)v.z | (uint32_t)v.w) > 255u);
      d4[t] = ((uint32_t)v.x & 255u) | (((uint32_t)v.y & 255u) << 8) | (((uint32_t)v.z & 255u) << 16) | ((uint32_t)v.w << 24);
    }
    if (lane < (count & 3)) { const int v = src[(nv << 2) + lane]; viol += ((uint32_t)v > 255u); dst[(nv << 2) + lane] = (uint8_t)v; }
  };
  stage(a.rowptr + (int64_t)N0 + g0, cr, srp);
  if (CHECK) stage(a.col + E0, ce, scol);
  wave_sync();
  if (lane < cnt) {
    const int n = n_l, e = e_l, rb = (np[lane] - N0) + lane, cb = (int)(ep[lane] - E0);
    if (n > 0 && (srp[rb] != 0 || srp[rb + n] != (uint8_t)e || e > 255)) ++viol;
    for (int u = 0; u < n; ++u) {
      const int rs = srp[rb + u], re = srp[rb + u + 1];
      if (re < rs || re > e) { ++viol; continue; }
      const int deg = re - rs;
      leaves += deg == 1;
      maxdeg = max(maxdeg, deg);
      if (CHECK) {
        int prev = -1;
        bool sorted = true;
        for (int j = rs; j < re; ++j) {
          const int x = scol[cb + j];
          if (x >= n || x == u) { ++viol; continue; }
          if (!sorted || x <= prev) {
            sorted = false;
            for (int jj = rs; jj < j; ++jj) if (scol[cb + jj] == x) { ++viol; break; }
          }
          prev = x;
          const int ws = srp[rb + x], we = srp[rb + x + 1];
          if (we < ws || we > e) { ++viol; continue; }
          bool found = false;
          for (int jj = ws; jj < we; ++jj) if (scol[cb + jj] == u) { found = true; break; }
          viol += !found;
        }
      }
    }
  }
}

template <bool CHECK>
__global__ void __launch_bounds__(256) csr_scan_kernel(const CsrScanArgs a) {
  extern __shared__ __align__(16) unsigned char scan_smem[];
  __shared__ int32_t s_np[4][68];
  __shared__ int64_t s_ep[4][66];
  __shared__ int32_t s_leaf[4][64];
  const int lane = lane_id(), w = wave_id();
  const int64_t g0l = ((int64_t)blockIdx.x * 4 + w) * 64;
  if (g0l >= a.G) return;                                   // (waves never meet at a workgroup barrier)
  const int g0 = (int)g0l, cnt = min(64, a.G - g0);
  int32_t *np = s_np[w];
  int64_t *ep = s_ep[w];
  int32_t *lf = s_leaf[w];
  if (lane < cnt) { np[lane] = a.node_ptr[g0 + lane]; ep[lane] = a.edge_ptr[g0 + lane]; }
  if (lane == 0) { np[cnt] = a.node_ptr[g0 + cnt]; ep[cnt] = a.edge_ptr[g0 + cnt]; }
  lf[lane] = 0;
  wave_sync();
  int viol = 0, maxdeg = 0, n_l = 0, leaves = 0;
  int64_t e_l = 0;
  if (lane < cnt) { n_l = np[lane + 1] - np[lane]; e_l = ep[lane + 1] - ep[lane]; }
  // nothing is indexed with a size that has not been looked at: a group with a negative count is flagged and skipped
  const bool sane = __ballot(lane < cnt && (n_l < 0 || e_l < 0 || e_l > 0x7FFFFFFF)) == 0;
  const bool staged = sane && a.cap_r > 0 && (int64_t)np[cnt] - np[0] + cnt <= a.cap_r && ep[cnt] - ep[0] <= (CHECK ? a.cap_e : 0x7FFFFFFF);
  if (!sane) {
    viol = lane == 0;
    n_l = 0; e_l = 0;
  } else if (staged) {
    unsigned char *mine = scan_smem + (size_t)w * (a.cap_r + (CHECK ? a.cap_e : 0));
    scan_group_staged<CHECK>(a, np, ep, g0, cnt, n_l, (int)e_l, mine, mine + a.cap_r, viol, maxdeg, leaves);
  } else {
    const int N0 = np[0], N1 = np[cnt];
    for (int v = N0 + lane; v < N1; v += kWave) {
      int lo = 0, hi = cnt - 1;                             // the largest k with np[k] <= v (empty graphs share a start: the last one owns v)
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (np[mid] <= v) lo = mid; else hi = mid - 1; }
      const int k = lo, u = v - np[k], n = np[k + 1] - np[k], eg = (int)(ep[k + 1] - ep[k]);
      const int32_t *__restrict__ rpg = a.rowptr + (int64_t)np[k] + g0 + k;     // the graph's n + 1 row pointers
      const int rs = rpg[u], re = rpg[u + 1];
      const bool bad = rs < 0 || re < rs || re > eg || (u == 0 && rs != 0) || (u == n - 1 && re != eg);
      const int deg = bad ? 0 : re - rs;
      viol += bad;
      maxdeg = max(maxdeg, deg);
      if (deg == 1) atomicAdd(&lf[k], 1);
      if (CHECK && !bad) {
        const int32_t *__restrict__ c = a.col + ep[k];
        int prev = -1;
        bool sorted = true;
        for (int j = rs; j < re; ++j) {
          const int x = c[j];
          if (x < 0 || x >= n || x == u) { ++viol; continue; }            // out of range, or a self loop
          if (!sorted || x <= prev) {                                     // rows usually ascend (then no entry repeats); else compare
            sorted = false;
            for (int jj = rs; jj < j; ++jj) if (c[jj] == x) { ++viol; break; }
          }
          prev = x;
          const int ws = rpg[x], we = rpg[x + 1];                         // (u, x) needs (x, u)
          if (ws < 0 || we < ws || we > eg) { ++viol; continue; }
          int b0 = ws, b1 = we;
          bool found = false;
          while (b0 < b1) {                                               // bisection is right when that row ascends ...
            const int mid = (b0 + b1) >> 1, y = c[mid];
            if (y == u) { found = true; break; }
            if (y < u) b0 = mid + 1; else b1 = mid;
          }
          if (!found)                                                     // ... and a scan settles it when it does not
            for (int jj = ws; jj < we; ++jj) if (c[jj] == u) { found = true; break; }
          viol += !found;
        }
      }
    }
  }
  wave_sync();
  if (a.keys && lane < cnt) a.keys[g0 + lane] = (uint8_t)min(255, n_l + leaves + lf[lane]);
  const int vs = (int)wave_sum64(viol), md = wave_max(maxdeg), mn = wave_max(n_l), me = wave_max((int)e_l);
  if (lane == 0) {
    // (four words of one line, 3.9 k waves: every wave's atomics in a row cost 130 us of ZINC-full's 144 - one word takes ~88
    // atomics per microsecond; a maximum that is already there needs none)
    if (vs) atomicAdd(a.info + kInfoViol, vs);
    raise_max(a.info + kInfoMaxDeg, md);
    raise_max(a.info + kInfoMaxN, mn);
    raise_max(a.info + kInfoMaxE, me);
  }
}

// launch of csr_scan_kernel: the staged path's LDS slice from what the caller says about the batch (claims: a group that does
// not fit them takes the general path, values that do not fit a byte are violations)
template <bool CHECK>
static void launch_csr_scan(CsrScanArgs a, const gtok_csr *g, hipStream_t s) {
  a.cap_r = a.cap_e = 0;
  if (g->max_nodes > 0 && g->max_nodes <= 255 && g->max_edges <= 255) {
    const int64_t cn = g->chunk_nodes > 0 ? g->chunk_nodes : 64ll * g->max_nodes;
    const int64_t ce = g->chunk_edges > 0 ? g->chunk_edges : 64ll * (g->max_edges > 0 ? g->max_edges : 1);
    const int64_t r = (cn + 64 + 15) / 16 * 16, e = (ce + 15) / 16 * 16;
    if (4 * (r + (CHECK ? e : 0)) <= 150 * 1024) { a.cap_r = (int)r; a.cap_e = (int)e; }
  }
  const size_t lds = (size_t)4 * (a.cap_r + (CHECK ? a.cap_e : 0));
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(csr_scan_kernel<CHECK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(csr_scan_kernel<CHECK>, dim3((unsigned)(((int64_t)a.G + 255) / 256)), dim3(256), lds, s, a);
}

// ---------------------------------------------------------------------------------------------------------------
// stable counting sort of the graphs by DESCENDING key (one byte per graph): block histograms -> one scan -> scatter
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSortBlock = 1024;      // keys per workgroup: wave w takes [256 w, 256 w + 256), 64 per step

__global__ void __launch_bounds__(256) lane_hist_kernel(const uint8_t *__restrict__ keys, int G, int nblocks, int32_t *__restrict__ hist) {
  __shared__ int s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortBlock;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < G) atomicAdd(&s_h[keys[i]], 1);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive scan of hist[bin][block] in the order (bin descending, block ascending), in place; one workgroup: the bins'
// totals (a wave per row, coalesced), their scan in descending bin order, then every row's own running sum
__global__ void __launch_bounds__(1024) lane_scan_kernel(int32_t *__restrict__ hist, int nblocks) {
  __shared__ int s_tot[256], s_base[256];
  const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
  for (int bin = w; bin < 256; bin += 16) {
    const int32_t *row = hist + (int64_t)bin * nblocks;
    int s = 0;
    for (int i = lane; i < nblocks; i += kWave) s += row[i];
    s = (int)wave_sum64(s);
    if (lane == 0) s_tot[bin] = s;
  }
  __syncthreads();
  if (w == 0) {                                     // base[bin] = sum of the totals of all HIGHER bins
    int carry = 0;
    for (int c = 0; c < 4; ++c) {
      const int bin = 255 - (c * kWave + lane);
      const int v = s_tot[bin];
      int inc = v;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
      s_base[bin] = carry + inc - v;
      carry += __shfl(inc, kWave - 1);
    }
  }
  __syncthreads();
  for (int bin = w; bin < 256; bin += 16) {
    if (s_tot[bin] == 0) continue;                  // (an empty bin's offsets are never read)
    int32_t *row = hist + (int64_t)bin * nblocks;
    int carry = s_base[bin];
    for (int i0 = 0; i0 < nblocks; i0 += kWave) {
      const int i = i0 + lane;
      const int v = i < nblocks ? row[i] : 0;
      int inc = v;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
      if (i < nblocks) row[i] = carry + inc - v;
      carry += __shfl(inc, kWave - 1);
    }
  }
}

__global__ void __launch_bounds__(256) lane_scatter_kernel(const uint8_t *__restrict__ keys, int G, int nblocks, const int32_t *__restrict__ offs,
                                                           const int32_t *__restrict__ node_ptr, const int64_t *__restrict__ edge_ptr,
                                                           int32_t *__restrict__ graph_ids, int32_t *__restrict__ n2, int32_t *__restrict__ e2) {
  __shared__ int s_off[256];
  __shared__ int s_tot[4][256], s_run[4][256];
  const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
  s_off[tid] = offs[(int64_t)tid * nblocks + blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_tot[k][tid] = 0; s_run[k][tid] = 0; }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortBlock + w * 256;
  int key[4];
  bool val[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * kWave + lane;
    val[j] = i < G;
    key[j] = val[j] ? keys[i] : 0;
    if (val[j]) atomicAdd(&s_tot[w][key[j]], 1);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // the lanes that hold this lane's key: one ballot per key bit
    uint64_t m = __ballot(val[j]);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (key[j] >> b) & 1;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int prior = s_run[w][key[j]];                       // equal keys the wave has placed in earlier steps
    wave_sync();
    const int rank = prior + __popcll(m & lanemask_lt());
    if (val[j] && lane == __builtin_ctzll(m)) s_run[w][key[j]] = prior + __popcll(m);
    wave_sync();
    int cross = 0;
    for (int k = 0; k < w; ++k) cross += s_tot[k][key[j]];    // equal keys of the workgroup's earlier waves
    if (val[j]) {
      const int64_t g = base + j * kWave + lane;
      const int dest = s_off[key[j]] + cross + rank;
      graph_ids[dest] = (int32_t)g;
      n2[dest] = node_ptr[g + 1] - node_ptr[g];
      e2[dest] = (int32_t)(edge_ptr[g + 1] - edge_ptr[g]);
    }
  }
}

// node_ptr2 / edge_ptr2 of the stored order: one pass, decoupled look-back over tiles of 1024 slots
constexpr int kPtrTile = 1024;
__global__ void __launch_bounds__(256) lane_ptr_scan_kernel(const int32_t *__restrict__ n2, const int32_t *__restrict__ e2, int G,
                                                            int64_t *__restrict__ st_n, int64_t *__restrict__ st_e, int *__restrict__ ticket,
                                                            int32_t *__restrict__ node_ptr2, int64_t *__restrict__ edge_ptr2) {
  __shared__ int s_tile;
  __shared__ int64_t s_wn[4], s_we[4], s_base[2];
  const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1);
  __syncthreads();
  const int tile = s_tile;
  const int64_t i0 = (int64_t)tile * kPtrTile + tid * 4;
  int nv[4], ev[4];
  int64_t tn = 0, te = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    nv[k] = i0 + k < G ? n2[i0 + k] : 0;
    ev[k] = i0 + k < G ? e2[i0 + k] : 0;
    tn += nv[k]; te += ev[k];
  }
  int64_t in_n = tn, in_e = te;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int64_t un = __shfl_up(in_n, o), ue = __shfl_up(in_e, o);
    if (lane >= o) { in_n += un; in_e += ue; }
  }
  if (lane == kWave - 1) { s_wn[w] = in_n; s_we[w] = in_e; }
  __syncthreads();
  if (w == 0) {
    const int64_t sn = s_wn[0] + s_wn[1] + s_wn[2] + s_wn[3], se = s_we[0] + s_we[1] + s_we[2] + s_we[3];
    const int64_t xn = lookback_exclusive(st_n, tile, sn), xe = lookback_exclusive(st_e, tile, se);
    if (lane == 0) { s_base[0] = xn; s_base[1] = xe; }
  }
  __syncthreads();
  int64_t rn = s_base[0] + in_n - tn, re = s_base[1] + in_e - te;
  for (int k = 0; k < w; ++k) { rn += s_wn[k]; re += s_we[k]; }
  if (tile == 0 && tid == 0) { node_ptr2[0] = 0; edge_ptr2[0] = 0; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rn += nv[k]; re += ev[k];
    if (i0 + k < G) { node_ptr2[i0 + k + 1] = (int32_t)rn; edge_ptr2[i0 + k + 1] = re; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the permuted CSR arrays (+ their byte mirror): one wave per 64 stored slots; destination ranges are contiguous, a
// destination element finds its slot by bisection over the group's offsets in LDS
// ---------------------------------------------------------------------------------------------------------------
struct GatherArgs {
  const int32_t *node_ptr; const int64_t *edge_ptr; const int32_t *rowptr; const int32_t *col; const uint8_t *nattr; const uint8_t *eattr;
  int G;
  const int32_t *graph_ids; const int32_t *node_ptr2; const int64_t *edge_ptr2;
  int32_t *rowptr2; int32_t *col2; uint8_t *nattr2; uint8_t *eattr2; uint8_t *rowptr8; uint8_t *col8;
};

constexpr int kGatherU = 8;
__global__ void __launch_bounds__(256) lane_gather_kernel(const GatherArgs a) {
  __shared__ int32_t s_dr[4][68], s_dn[4][68], s_de[4][68], s_g[4][64], s_sn[4][64];
  __shared__ int64_t s_se[4][64];
  const int lane = lane_id(), w = wave_id();
  const int64_t s0l = ((int64_t)blockIdx.x * 4 + w) * 64;
  if (s0l >= a.G) return;
  const int s0 = (int)s0l, cnt = min(64, a.G - s0);
  int32_t *dr = s_dr[w], *dn = s_dn[w], *de = s_de[w], *gi = s_g[w], *sn = s_sn[w];
  int64_t *se = s_se[w];
  const int64_t E0 = a.edge_ptr2[s0];
  const int N0 = a.node_ptr2[s0];
  if (lane < cnt) {
    const int g = a.graph_ids[s0 + lane];
    gi[lane] = g; sn[lane] = a.node_ptr[g]; se[lane] = a.edge_ptr[g];
    const int d = a.node_ptr2[s0 + lane] - N0;
    dn[lane] = d; dr[lane] = d + lane; de[lane] = (int)(a.edge_ptr2[s0 + lane] - E0);
  }
  if (lane == 0) { const int d = a.node_ptr2[s0 + cnt] - N0; dn[cnt] = d; dr[cnt] = d + cnt; de[cnt] = (int)(a.edge_ptr2[s0 + cnt] - E0); }
  wave_sync();
  const int Nn = dn[cnt], Ne = de[cnt];
  // A lane's elements t = lane, lane + 64, ... ascend, so the slot that owns t only ever moves forward: `own` advances a
  // running index over the group's offsets (1-3 LDS reads per element instead of a 6-step bisection whose result the
  // element's load then waits for); eight elements per pass, their loads in flight together (a scattered load
  // takes ~2 us under this load: the passes, not the bytes, are the kernel's time).
  auto own = [](const int32_t *start, int cnt_, int t, int &k) { while (k + 1 < cnt_ && start[k + 1] <= t) ++k; };
  {   // row pointers: slot k's n_k + 1 values start at dn[k] + k
    int k = 0;
    const int total = Nn + cnt;
    for (int t0 = lane; t0 < total; t0 += kGatherU * kWave) {
      int v[kGatherU]; int64_t at[kGatherU]; bool on[kGatherU];
#pragma unroll
      for (int j = 0; j < kGatherU; ++j) {
        const int t = t0 + j * kWave;
        on[j] = t < total;
        if (on[j]) { own(dr, cnt, t, k); v[j] = a.rowptr[(int64_t)sn[k] + gi[k] + (t - dr[k])]; at[j] = (int64_t)N0 + s0 + t; }
      }
#pragma unroll
      for (int j = 0; j < kGatherU; ++j)
        if (on[j]) { a.rowptr2[at[j]] = v[j]; if (a.rowptr8) a.rowptr8[at[j]] = (uint8_t)v[j]; }
    }
  }
  if (a.nattr) {
    int k = 0;
    for (int t0 = lane; t0 < Nn; t0 += kGatherU * kWave) {
      uint8_t v[kGatherU]; bool on[kGatherU];
#pragma unroll
      for (int j = 0; j < kGatherU; ++j) {
        const int t = t0 + j * kWave;
        on[j] = t < Nn;
        if (on[j]) { own(dn, cnt, t, k); v[j] = a.nattr[(int64_t)sn[k] + (t - dn[k])]; }
      }
#pragma unroll
      for (int j = 0; j < kGatherU; ++j) if (on[j]) a.nattr2[(int64_t)N0 + t0 + j * kWave] = v[j];
    }
  }
  {
    int k = 0;
    for (int t0 = lane; t0 < Ne; t0 += kGatherU * kWave) {
      int v[kGatherU]; uint8_t ev[kGatherU]; bool on[kGatherU];
#pragma unroll
      for (int j = 0; j < kGatherU; ++j) {
        const int t = t0 + j * kWave;
        on[j] = t < Ne;
        if (on[j]) {
          own(de, cnt, t, k);
          const int64_t src = se[k] + (t - de[k]);
          v[j] = a.col[src];
          if (a.eattr) ev[j] = a.eattr[src];
        }
      }
#pragma unroll
      for (int j = 0; j < kGatherU; ++j) {
        if (!on[j]) continue;
        const int64_t at = E0 + t0 + j * kWave;
        a.col2[at] = v[j];
        if (a.col8) a.col8[at] = (uint8_t)v[j];
        if (a.eattr) a.eattr2[at] = ev[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// units: greedy over the stored order - a unit takes slots while it has < 64 of them and their node and entry sums stay
// within the caps (the LDS budget split between the two in the corpus' own proportion).  next(i) is independent per slot
// (one bisection each); the chain 0, next(0), next(next(0)), ... is followed block-wise: every block of 2048 slots
// tabulates, for each of the 64 offsets a chain can enter it at, where that chain leaves it and how many units it cut;
// one thread then hops from block to block through the tables (in LDS), and the blocks write their unit starts.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kUnitBlock = 2048, kChaseChunk = 128;

__global__ void __launch_bounds__(256) unit_hop_kernel(const int32_t *__restrict__ cn, const int64_t *__restrict__ ce, int G, int budget,
                                                       uint8_t *__restrict__ hop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G) return;
  // (double arithmetic as the Python restatement in tests does it: a division and an add, nothing to contract)
  const int room = (budget - (64 + 16 + 8 + 8 + 8) - 4 * 15) / 2;
  const double nsum = (double)cn[G], esum = (double)ce[G];
  const double ratio = esum / (nsum > 1.0 ? nsum : 1.0);
  const int ncap = max(64, (int)((double)room / (1.0 + ratio)));
  const int ecap = max(255, room - ncap);
  const int64_t nlim = (int64_t)cn[i] + ncap, elim = ce[i] + ecap;
  int64_t lo = i, hi = min((int64_t)G, i + 64);             // the largest j in [i, i + 64] with both sums within the caps
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if ((int64_t)cn[mid] <= nlim && ce[mid] <= elim) lo = mid; else hi = mid - 1;
  }
  hop[i] = (uint8_t)(max(lo, i + 1) - i);                   // (a graph too large for the caps is a unit of its own)
}

__device__ __forceinline__ void stage_hops(const uint8_t *__restrict__ hop, int64_t base, uint8_t *s_h) {
  // (the hop array is padded to whole blocks: 16-byte loads; values beyond G are never followed)
  const U8x16 *src = reinterpret_cast<const U8x16 *>(hop + base);
  U8x16a *dst = reinterpret_cast<U8x16a *>(s_h);
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < kUnitBlock / 16 / kWave; ++k) { const U8x16 x = src[lane + k * kWave]; dst[lane + k * kWave] = U8x16a{x.a, x.b, x.c, x.d}; }
  wave_sync();
}

__global__ void __launch_bounds__(64) unit_block_kernel(const uint8_t *__restrict__ hop, int G, uint32_t *__restrict__ table) {
  __shared__ __align__(16) uint8_t s_h[kUnitBlock];
  const int64_t base = (int64_t)blockIdx.x * kUnitBlock, end = min((int64_t)G, base + kUnitBlock);
  stage_hops(hop, base, s_h);
  const int lane = lane_id();
  int64_t pos = base + lane;
  int cnt = 0;
  while (pos < end) { pos += s_h[pos - base]; ++cnt; }
  const int out = cnt ? (int)min((int64_t)63, pos - end) : 0;
  table[(int64_t)blockIdx.x * 64 + lane] = (uint32_t)out | ((uint32_t)cnt << 8);
}

__global__ void __launch_bounds__(256) unit_chase_kernel(const uint32_t *__restrict__ table, int nb, int G, int32_t *__restrict__ entry,
                                                         int32_t *__restrict__ ubase, int32_t *__restrict__ unit_ptr, int32_t *__restrict__ info) {
  __shared__ uint32_t s_t[kChaseChunk * 64];
  int e = 0, base = 0;
  for (int c0 = 0; c0 < nb; c0 += kChaseChunk) {
    const int nbc = min(kChaseChunk, nb - c0);
    for (int i = (int)threadIdx.x; i < nbc * 64; i += 256) s_t[i] = table[(int64_t)c0 * 64 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int b = 0; b < nbc; ++b) {
        const uint32_t t = s_t[b * 64 + e];
        entry[c0 + b] = e; ubase[c0 + b] = base;
        base += (int)(t >> 8);
        e = (int)(t & 255u);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { info[kInfoUnits] = base; unit_ptr[base] = G; }
}

__global__ void __launch_bounds__(64) unit_write_kernel(const uint8_t *__restrict__ hop, int G, const int32_t *__restrict__ entry,
                                                        const int32_t *__restrict__ ubase, int32_t *__restrict__ unit_ptr) {
  __shared__ __align__(16) uint8_t s_h[kUnitBlock];
  const int64_t base = (int64_t)blockIdx.x * kUnitBlock, end = min((int64_t)G, base + kUnitBlock);
  stage_hops(hop, base, s_h);
  if (lane_id() == 0) {
    int64_t pos = base + entry[blockIdx.x];
    int u = ubase[blockIdx.x];
    while (pos < end) { unit_ptr[u++] = (int32_t)pos; pos += s_h[pos - base]; }
  }
}

__global__ void __launch_bounds__(256) unit_info_kernel(const int32_t *__restrict__ unit_ptr, const int32_t *__restrict__ cn, const int64_t *__restrict__ ce,
                                                        int32_t *__restrict__ unit_info, int32_t *__restrict__ info) {
  const int nu = info[kInfoUnits];
  if ((int64_t)blockIdx.x * 256 >= nu) return;
  const int u = (int)blockIdx.x * 256 + (int)threadIdx.x;
  int ns = 0, es = 0;
  if (u < nu) {
    const int s0 = unit_ptr[u], s1 = unit_ptr[u + 1];
    const int n0 = cn[s0], n1 = cn[s1];
    const int64_t e0 = ce[s0], e1 = ce[s1];
    int32_t *r = unit_info + 8 * (int64_t)u;
    r[0] = s0; r[1] = s1; r[2] = n0; r[3] = n1;
    r[4] = (int32_t)(uint32_t)e0; r[5] = (int32_t)(uint32_t)((uint64_t)e0 >> 32);
    r[6] = (int32_t)(uint32_t)e1; r[7] = (int32_t)(uint32_t)((uint64_t)e1 >> 32);
    ns = n1 - n0; es = (int)(e1 - e0);
  }
  ns = wave_max(ns); es = wave_max(es);
  if (lane_id() == 0) { raise_max(info + kInfoChunkN, ns); raise_max(info + kInfoChunkE, es); }
}

__global__ void __launch_bounds__(256) csr_ws_init_kernel(int64_t *__restrict__ states, int64_t nstates, int *__restrict__ ticket, int32_t *__restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nstates) states[i] = kTileEmpty;
  if (i == 0) *ticket = 0;
  if (i < 8) info[i] = 0;
}

// workspace layout of gtok_csr_lane_sort (every region 16-byte aligned)
struct LaneSortWs {
  int64_t keys, hop, n2, e2, hist, st_n, st_e, ticket, table, entry, ubase, total;
  int nblocks, tiles, ublocks;
};
static LaneSortWs lane_sort_ws(int64_t G) {
  LaneSortWs w;
  auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
  w.nblocks = (int)((G + kSortBlock - 1) / kSortBlock);
  w.tiles = (int)((G + kPtrTile - 1) / kPtrTile);
  w.ublocks = (int)((G + kUnitBlock - 1) / kUnitBlock);
  int64_t off = 0;
  w.keys = off; off += up(G);
  w.hop = off; off += up((int64_t)w.ublocks * kUnitBlock);
  w.n2 = off; off += up(4 * G);
  w.e2 = off; off += up(4 * G);
  w.hist = off; off += up(4 * 256ll * w.nblocks);
  w.st_n = off; off += up(8ll * w.tiles);
  w.st_e = off; off += up(8ll * w.tiles);       // (st_n and st_e are one run of 2 * tiles words for the initialiser)
  w.ticket = off; off += 16;
  w.table = off; off += up(4ll * 64 * w.ublocks);
  w.entry = off; off += up(4ll * w.ublocks);
  w.ubase = off; off += up(4ll * w.ublocks);
  w.total = off;
  return w;
}

}  // namespace gtok

using namespace gtok;

extern "C" int gtok_csr_check(const gtok_csr *g, int32_t *info, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!g || !info || g->num_graphs < 0 || g->graph_ids || g->unit_ptr || (g->max_edges > 0 && !g->col)) return GTOK_E_INVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(info, 0, 8 * sizeof(int32_t), s) != hipSuccess) return GTOK_E_LAUNCH;
  if (g->num_graphs == 0) return GTOK_OK;
  if (!g->node_ptr || !g->edge_ptr || !g->rowptr) return GTOK_E_INVAL;      // (col may be NULL only when no graph has an entry: the kernel then reads none)
  CsrScanArgs a;
  a.node_ptr = g->node_ptr; a.edge_ptr = g->edge_ptr; a.rowptr = g->rowptr; a.col = g->col; a.G = g->num_graphs; a.info = info; a.keys = nullptr;
  if (g->col) launch_csr_scan<true>(a, g, s);
  else launch_csr_scan<false>(a, g, s);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int64_t gtok_csr_lane_sort_workspace(int32_t num_graphs) {
  return num_graphs <= 0 ? 16 : lane_sort_ws(num_graphs).total;
}

extern "C" int gtok_csr_lane_sort(const gtok_csr *g, int32_t lds_budget, int32_t check, const gtok_csr_sorted *o, void *workspace,
                                  int64_t workspace_bytes, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!g || !o || g->num_graphs < 0 || g->graph_ids || g->unit_ptr || lds_budget < 0) return GTOK_E_INVAL;
  if (g->max_nodes > 64 || g->max_edges > 255) return GTOK_E_TOO_LARGE;      // what the lane-per-graph kernel takes
  if (!o->info) return GTOK_E_INVAL;
  hipStream_t s = (hipStream_t)stream;
  const int G = g->num_graphs;
  if (G == 0) return hipMemsetAsync(o->info, 0, 8 * sizeof(int32_t), s) == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
  if (!g->node_ptr || !g->edge_ptr || !g->rowptr || (g->max_edges > 0 && !g->col)) return GTOK_E_INVAL;
  if (!o->graph_ids || !o->node_ptr || !o->edge_ptr || !o->rowptr || (g->max_edges > 0 && !o->col) || !o->unit_ptr || !o->unit_info)
    return GTOK_E_INVAL;
  if ((g->nattr && !o->nattr) || (g->eattr && !o->eattr)) return GTOK_E_INVAL;
  const LaneSortWs w = lane_sort_ws(G);
  if (!workspace || workspace_bytes < w.total || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return GTOK_E_INVAL;
  const int budget = lds_budget > 0 ? lds_budget : 10224;      // 160 KB / 16 waves, less the workgroup's shared words (gtok_sent.hip)
  if (budget < 512) return GTOK_E_INVAL;
  uint8_t *ws = reinterpret_cast<uint8_t *>(workspace);
  uint8_t *keys = ws + w.keys, *hop = ws + w.hop;
  int32_t *n2 = reinterpret_cast<int32_t *>(ws + w.n2), *e2 = reinterpret_cast<int32_t *>(ws + w.e2), *hist = reinterpret_cast<int32_t *>(ws + w.hist);
  int64_t *st_n = reinterpret_cast<int64_t *>(ws + w.st_n), *st_e = reinterpret_cast<int64_t *>(ws + w.st_e);
  int *ticket = reinterpret_cast<int *>(ws + w.ticket);
  uint32_t *table = reinterpret_cast<uint32_t *>(ws + w.table);
  int32_t *entry = reinterpret_cast<int32_t *>(ws + w.entry), *ubase = reinterpret_cast<int32_t *>(ws + w.ubase);

  const int64_t nstates = (w.st_e - w.st_n) / 8 + w.tiles;     // both status arrays in one run
  hipLaunchKernelGGL(csr_ws_init_kernel, dim3((unsigned)((nstates + 255) / 256 > 0 ? (nstates + 255) / 256 : 1)), dim3(256), 0, s, st_n, nstates, ticket, o->info);
  CsrScanArgs a;
  a.node_ptr = g->node_ptr; a.edge_ptr = g->edge_ptr; a.rowptr = g->rowptr; a.col = g->col; a.G = G; a.info = o->info; a.keys = keys;
  const unsigned ngroups = (unsigned)(((int64_t)G + 255) / 256);
  if (check && g->col) launch_csr_scan<true>(a, g, s);
  else launch_csr_scan<false>(a, g, s);
  hipLaunchKernelGGL(lane_hist_kernel, dim3(w.nblocks), dim3(256), 0, s, keys, G, w.nblocks, hist);
  hipLaunchKernelGGL(lane_scan_kernel, dim3(1), dim3(1024), 0, s, hist, w.nblocks);
  hipLaunchKernelGGL(lane_scatter_kernel, dim3(w.nblocks), dim3(256), 0, s, keys, G, w.nblocks, hist, g->node_ptr, g->edge_ptr, o->graph_ids, n2, e2);
  hipLaunchKernelGGL(lane_ptr_scan_kernel, dim3(w.tiles), dim3(256), 0, s, n2, e2, G, st_n, st_e, ticket, o->node_ptr, o->edge_ptr);
  GatherArgs ga;
  ga.node_ptr = g->node_ptr; ga.edge_ptr = g->edge_ptr; ga.rowptr = g->rowptr; ga.col = g->col; ga.nattr = g->nattr; ga.eattr = g->eattr; ga.G = G;
  ga.graph_ids = o->graph_ids; ga.node_ptr2 = o->node_ptr; ga.edge_ptr2 = o->edge_ptr;
  ga.rowptr2 = o->rowptr; ga.col2 = o->col; ga.nattr2 = o->nattr; ga.eattr2 = o->eattr; ga.rowptr8 = o->rowptr8; ga.col8 = o->col8;
  hipLaunchKernelGGL(lane_gather_kernel, dim3(ngroups), dim3(256), 0, s, ga);
  hipLaunchKernelGGL(unit_hop_kernel, dim3(ngroups), dim3(256), 0, s, o->node_ptr, o->edge_ptr, G, budget, hop);
  hipLaunchKernelGGL(unit_block_kernel, dim3(w.ublocks), dim3(64), 0, s, hop, G, table);
  hipLaunchKernelGGL(unit_chase_kernel, dim3(1), dim3(256), 0, s, table, w.ublocks, G, entry, ubase, o->unit_ptr, o->info);
  hipLaunchKernelGGL(unit_write_kernel, dim3(w.ublocks), dim3(64), 0, s, hop, G, entry, ubase, o->unit_ptr);
  hipLaunchKernelGGL(unit_info_kernel, dim3(ngroups), dim3(256), 0, s, o->unit_ptr, o->node_ptr, o->edge_ptr, o->unit_info, o->info);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}
