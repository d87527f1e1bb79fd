// gtok_rows.hip — the PACKED (ragged) row format of token slabs: offsets, pack, unpack, collate.
//
// A tokenizer entry point writes a padded [rows, ld] int32 slab (include/gtok.h).  More than half of a ZINC slab is
// padding and every id fits 16 bits, so the copies that leave the GPU - the RCCL all-gather of BASELINE config 4
// (trainer/train_agtt.py:602-607 needs the rows in dataset order on every rank) and the D2H copy behind
// TokenizedGraphDataset.__getitem__ (trainer/train_agtt.py:246-273) - move the packed form: row r's
// n_r = min(len[r], ld) ids, 16 or 32 bits each, contiguous from element row_ptr[r].  Pure data movement: HBM-bound,
// no LDS, every access a 16-byte vector when the row starts allow it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtok.h"
#include "gtok_common.hpp"

namespace gtok {

constexpr int kScanBlock = 256, kScanItems = 16, kScanTile = kScanBlock * kScanItems;   // 4096 rows per scan block

__device__ __forceinline__ int64_t row_cost(const int32_t *__restrict__ len, int64_t i, int ld, int align_mask) {
  int n = len[i];
  n = n < 0 ? 0 : (n > ld ? ld : n);
  return (int64_t)((n + align_mask) & ~align_mask);
}

// block-wide exclusive scan of one int64 per thread (kScanBlock threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int64_t block_exclusive(int64_t v, int64_t *total) {
  __shared__ int64_t wave_sum[kScanBlock / kWave];
  const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
  int64_t inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int64_t up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == kWave - 1) wave_sum[w] = inc;
  __syncthreads();
  int64_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kScanBlock / kWave; ++k) {
    const int64_t s = wave_sum[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// pass A: the sum of every tile of kScanTile rows lands in the row_ptr slot that closes the tile
__global__ void __launch_bounds__(kScanBlock) row_tile_sums_kernel(const int32_t *__restrict__ len, int64_t rows, int ld,
                                                                   int align_mask, int64_t *__restrict__ row_ptr) {
  const int64_t lo = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (lo + k < rows) s += row_cost(len, lo + k, ld, align_mask);
  int64_t total;
  (void)block_exclusive(s, &total);
  if (threadIdx.x == 0) {
    const int64_t end = ((int64_t)blockIdx.x + 1) * kScanTile;
    row_ptr[end < rows ? end : rows] = total;
  }
}

// pass B: one workgroup turns the tile sums into running totals, in place (tiles <= rows / 4096: a few hundred)
__global__ void __launch_bounds__(kScanBlock) row_tile_scan_kernel(int64_t rows, int64_t *__restrict__ row_ptr) {
  const int64_t tiles = (rows + kScanTile - 1) / kScanTile;
  int64_t carry = 0;
  for (int64_t t0 = 0; t0 < tiles; t0 += kScanBlock) {
    const int64_t t = t0 + threadIdx.x;
    const int64_t end = (t + 1) * kScanTile;
    const int64_t slot = end < rows ? end : rows;
    const int64_t v = t < tiles ? row_ptr[slot] : 0;
    int64_t total;
    const int64_t ex = block_exclusive(v, &total);
    if (t < tiles) row_ptr[slot] = carry + ex + v;
    carry += total;
  }
  if (threadIdx.x == 0) row_ptr[0] = 0;
}

// pass C: every tile fills the slots strictly inside it from its base (the slot that closes the previous tile)
__global__ void __launch_bounds__(kScanBlock) row_tile_fill_kernel(const int32_t *__restrict__ len, int64_t rows, int ld,
                                                                   int align_mask, int64_t *__restrict__ row_ptr) {
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  const int64_t lo = tile0 + (int64_t)threadIdx.x * kScanItems;
  int64_t c[kScanItems];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    c[k] = lo + k < rows ? row_cost(len, lo + k, ld, align_mask) : 0;
    s += c[k];
  }
  int64_t total;
  int64_t run = block_exclusive(s, &total) + row_ptr[tile0];
  const int64_t tile_end = tile0 + kScanTile;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    run += c[k];
    const int64_t slot = lo + k + 1;                      // row_ptr[slot] = end of row slot - 1
    if (slot < rows && slot < tile_end) row_ptr[slot] = run;
  }
}

// a row's start in the packed buffer: segments (the blocks the ranks of an all-gather contributed) restart at
// multiples of segment_stride elements
__device__ __forceinline__ int64_t packed_start(const int64_t *__restrict__ row_ptr, int64_t r, int segment_rows,
                                                int64_t segment_stride, int ld, int at = 0) {
  if (!row_ptr) return r * (int64_t)ld;            // the strided form: a [rows, ld] slab (GTOK_SENT_U16) read in place
  if (at) {                                        // explicit starts, relative to the row's segment (gtok_unpack_rows_at)
    const int64_t st = row_ptr[r];
    return st < 0 ? (int64_t)-1 : (segment_rows > 0 ? (r / segment_rows) * segment_stride : (int64_t)0) + st;
  }
  if (segment_rows <= 0) return row_ptr[r];
  const int64_t seg = r / segment_rows;
  return seg * segment_stride + (row_ptr[r] - row_ptr[seg * (int64_t)segment_rows]);
}

struct RowsArgs {
  const void *ids;         // pack: source slab (int32, or uint16 for gtok_pack_rows_u16); unpack: unused
  void *out_ids;           // unpack: destination slab (int32, or 16-bit ids: gtok_unpack_rows_u16)
  const int32_t *len;
  const int64_t *row_ptr;
  void *packed;
  int32_t *status;
  int64_t rows;
  int ld, pad_id, segment_rows, tpr_shift;   // threads per row = 1 << tpr_shift
  int at;                                    // unpack: row_ptr holds explicit row starts relative to the row's segment
  int64_t segment_stride, capacity;   // pack: elements `packed` can hold; unpack: elements it holds (0 = unknown)
};

// S = source id type (int32_t slab, or uint16_t: a GTOK_SENT_U16 slab), E = packed id type (uint16_t, int32_t or
// int64_t).  A thread owns pieces of 8 consecutive ids of one row: 16-byte loads and stores; 1 << tpr_shift threads
// share a row, 256 >> tpr_shift rows a workgroup.
template <typename S, typename E>
__device__ __forceinline__ uint32_t copy_row_pieces(const S *__restrict__ row, E *__restrict__ dst, int n, int sub, int tpr) {
  const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
  uint32_t wide = 0;
  for (int i = sub * 8; i < n; i += tpr * 8) {
    if (vec && i + 8 <= n) {
      uint32_t t[8];
      if (sizeof(S) == 4) {
        const int4 lo = *reinterpret_cast<const int4 *>(row + i), hi = *reinterpret_cast<const int4 *>(row + i + 4);
        t[0] = (uint32_t)lo.x; t[1] = (uint32_t)lo.y; t[2] = (uint32_t)lo.z; t[3] = (uint32_t)lo.w;
        t[4] = (uint32_t)hi.x; t[5] = (uint32_t)hi.y; t[6] = (uint32_t)hi.z; t[7] = (uint32_t)hi.w;
        wide |= t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7];
      } else {
        const uint4 p = *reinterpret_cast<const uint4 *>(row + i);
        t[0] = p.x & 0xFFFFu; t[1] = p.x >> 16; t[2] = p.y & 0xFFFFu; t[3] = p.y >> 16;
        t[4] = p.z & 0xFFFFu; t[5] = p.z >> 16; t[6] = p.w & 0xFFFFu; t[7] = p.w >> 16;
      }
      if (sizeof(E) == 2) {
        uint4 o;
        if (sizeof(S) == 2) {
          o = *reinterpret_cast<const uint4 *>(row + i);
        } else {
          o.x = (t[0] & 0xFFFFu) | (t[1] << 16); o.y = (t[2] & 0xFFFFu) | (t[3] << 16);
          o.z = (t[4] & 0xFFFFu) | (t[5] << 16); o.w = (t[6] & 0xFFFFu) | (t[7] << 16);
        }
        *reinterpret_cast<uint4 *>(dst + i) = o;
      } else if (sizeof(E) == 4) {
        *reinterpret_cast<uint4 *>(dst + i) = make_uint4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<uint4 *>(dst + i + 4) = make_uint4(t[4], t[5], t[6], t[7]);
      } else {                                        // int64 ids: ids are never negative on this path (a slab of 16-bit ids, or sign-extended int32)
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const int64_t a0 = sizeof(S) == 4 ? (int64_t)(int32_t)t[k] : (int64_t)t[k], a1 = sizeof(S) == 4 ? (int64_t)(int32_t)t[k + 1] : (int64_t)t[k + 1];
          *reinterpret_cast<longlong2 *>(dst + i + k) = longlong2{a0, a1};
        }
      }
    } else {
      for (int k = i; k < n && k < i + 8; ++k) {
        const S t = row[k];
        if (sizeof(S) == 4) wide |= (uint32_t)t;
        dst[k] = (E)t;
      }
    }
  }
  return wide;
}

template <typename S, typename E>
__global__ void __launch_bounds__(256) pack_rows_kernel(const RowsArgs a) {
  const int tpr = 1 << a.tpr_shift, sub = (int)threadIdx.x & (tpr - 1);
  const int64_t r = (int64_t)blockIdx.x * (256 >> a.tpr_shift) + ((int)threadIdx.x >> a.tpr_shift);
  if (r >= a.rows) return;
  int n = a.len[r];
  n = n < 0 ? 0 : (n > a.ld ? a.ld : n);
  const int64_t start = packed_start(a.row_ptr, r, a.segment_rows, a.segment_stride, a.ld);
  const S *__restrict__ row = reinterpret_cast<const S *>(a.ids) + r * (int64_t)a.ld;
  if (start + n > a.capacity) {                      // a caller-sized buffer that turned out too small: skip, flag
    if (sub == 0) atomicOr(a.status, 2);
    return;
  }
  const uint32_t wide = copy_row_pieces<S, E>(row, reinterpret_cast<E *>(a.packed) + start, n, sub, tpr);
  if (sizeof(E) == 2 && sizeof(S) == 4 && (wide & 0xFFFF0000u)) atomicOr(a.status, 1);     // an id that does not fit 16 bits
}

// gtok_row_offsets + gtok_pack_rows(_u16) in ONE launch (gtok_pack_rows_scan): a workgroup takes a tile of 256 rows, sums
// their packed sizes, learns where the tile starts from the tiles before it (decoupled look-back: the status word of tile t
// IS the row_ptr slot that closes it, which ends up holding exactly the tile's inclusive prefix), writes its row_ptr
// entries and copies its rows.  The three scan launches and the second read of the lengths are gone; what remains is the
// one read of the ids and the one write of the packed form.
constexpr int kPackU = 4;
struct PackScanArgs {
  const void *ids; const int32_t *len; int64_t *row_ptr; void *packed; int32_t *status; int *ticket;
  int64_t rows, capacity;
  int ld, align_mask, tpr_shift, tiles, chunk;      // chunk: consecutive tiles per workgroup (one ticket each)
  int fast;       // 8-id alignment, every slab row and the packed buffer on 16-byte boundaries: the flat piece loop
};

__global__ void __launch_bounds__(256) pack_scan_init_kernel(int64_t *__restrict__ row_ptr, int64_t rows, int tiles, int tile_rows, int32_t *__restrict__ status) {
  const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (t < tiles) { const int64_t end = ((int64_t)t + 1) * tile_rows; row_ptr[end < rows ? end : rows] = kTileEmpty; }
  if (t == 0) { row_ptr[0] = 0; *status = 0; }
}

// TILE rows per tile (64, 128 or 256), 256 threads; a workgroup takes a.chunk consecutive tiles on ONE ticket (a device-wide
// counter hands out ~88 tickets per microsecond: with a ticket per tile the 15.6 k tiles of 16 ZINC-full epochs waited 170 us
// in line).  Tiles are numbered in ticket order, so every lower tile belongs to a workgroup that is running or done - whatever
// order HIP dispatches workgroups in.  The sums of ALL the chunk's tiles are published before anything else is done (a
// successor must never wait for this workgroup's copies), then ONE look-back gives the chunk's start and with it every
// tile's inclusive prefix.
constexpr int kPackChunkMax = 8;
template <typename S, typename E, int TILE>
__global__ void __launch_bounds__(256) pack_scan_kernel(const PackScanArgs a) {
  constexpr int kPackTile = TILE, kScanWaves = TILE / kWave;
  __shared__ int s_tile;
  __shared__ int64_t s_wsum[4], s_base, s_start[kPackTile], s_tsum[kPackChunkMax];
  __shared__ int s_n[kPackTile], s_p[kPackTile], s_len[kPackChunkMax][kPackTile], s_part[kPackChunkMax][4];
  const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid == 0) {
    const int t = atomicAdd(a.ticket, 1);
    if (t == (a.tiles + a.chunk - 1) / a.chunk - 1) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last draw re-arms the counter
    s_tile = t * a.chunk;
  }
  __syncthreads();
  const int tile_first = s_tile, ntile = min(a.tiles - tile_first, a.chunk);
  int64_t *const rp = a.row_ptr;
  const int64_t rows = a.rows;
  auto word = [rp, rows](int t) { const int64_t end = ((int64_t)t + 1) * kPackTile; return rp + (end < rows ? end : rows); };
  // ---- phase 1: the lengths and sums of the chunk's tiles; the sums are published at once
  for (int i = 0; i < ntile; ++i) {
    const int64_t r = (int64_t)(tile_first + i) * kPackTile + tid;
    int c = 0;
    if (tid < kPackTile) {
      int n = 0;
      if (r < rows) { n = a.len[r]; n = n < 0 ? 0 : (n > a.ld ? a.ld : n); c = (n + a.align_mask) & ~a.align_mask; }
      s_len[i][tid] = n;
    }
    if (w < kScanWaves) {
      const int ws = (int)wave_sum64(c);
      if (lane == 0) s_part[i][w] = ws;
    }
  }
  __syncthreads();
  if (w == 0) {
    int64_t mysum = 0;
    if (lane < ntile) {
      for (int k = 0; k < kScanWaves; ++k) mysum += s_part[lane][k];
      s_tsum[lane] = mysum;
      if (lane > 0 || tile_first > 0)      // (tile 0's word goes straight to its prefix below)
        __hip_atomic_store(word(tile_first + lane), -mysum - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int64_t first_sum = __shfl(mysum, 0);
    const int64_t x = lookback_exclusive(word, tile_first, first_sum);       // (publishes the first tile's prefix itself)
    // every later tile of the chunk: its inclusive prefix follows from the chunk's start
    int64_t inc = mysum;
#pragma unroll
    for (int o = 1; o < kPackChunkMax; o <<= 1) { const int64_t up = __shfl_up(inc, o); if (lane >= o) inc += up; }
    if (lane > 0 && lane < ntile) __hip_atomic_store(word(tile_first + lane), x + inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) s_base = x;
  }
  __syncthreads();
  int64_t chunk_base = s_base;
  // ---- phase 2: tile by tile - row offsets, row_ptr, the copy
  for (int i = 0; i < ntile; ++i) {
  const int tile = tile_first + i;
  const int64_t r = (int64_t)tile * kPackTile + tid;
  const bool mine = tid < kPackTile && r < rows;
  const int n = tid < kPackTile ? s_len[i][tid] : 0;
  const int64_t cost = mine ? (int64_t)((n + a.align_mask) & ~a.align_mask) : 0;
  int64_t inc = cost;
  if (w < kScanWaves) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const int64_t up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  }
  int64_t start = chunk_base + inc - cost;
  for (int k = 0; k < w && k < kScanWaves; ++k) start += s_part[i][k];
  if (tid < kPackTile) { s_start[tid] = start; s_n[tid] = n; }
  // row_ptr[r + 1] = the end of row r; the slot that closes the tile is its status word (it holds the same value)
  if (mine && tid != kPackTile - 1 && r + 1 < rows) a.row_ptr[r + 1] = start + cost;
  const int64_t tbase_tile = chunk_base;
  const int64_t tile_sum = s_tsum[i];
  chunk_base += tile_sum;
  uint32_t wide = 0;
  bool over = false;
  if (a.fast) {
    // 8-id pieces, flat over the tile: with 8-id alignment the tile's packed region is ONE run of whole 16-byte (E = 2 bytes)
    // pieces, piece j of the tile landing at tile_start + 8 j.  A thread takes pieces tid, tid + 256, ... - kPackU of them per
    // pass, their loads in flight together (a row-per-thread-group loop had one load in flight per thread: 61 us for ZINC-full's
    // 44 MB in + 44 MB out, slower than the four launches it replaced) - and finds a piece's row by bisection over the rows'
    // piece offsets in LDS.  Tail pieces are loaded whole (a slab row is a multiple of 8 ids wide) and masked to zeros.
    const int64_t tbase = tbase_tile;
    if (tid < kPackTile) s_p[tid] = (int)((start - tbase) >> 3);
    __syncthreads();
    const int total_p = (int)(tile_sum >> 3);
    const int64_t row0 = (int64_t)tile * kPackTile;
    const int nrows = (int)min((int64_t)kPackTile, a.rows - row0);
    for (int j0 = tid; j0 < total_p; j0 += 256 * kPackU) {
      uint4 lo[kPackU], hi[kPackU];
      int cnt[kPackU];
#pragma unroll
      for (int u = 0; u < kPackU; ++u) {
        const int j = j0 + u * 256;
        cnt[u] = 0;
        if (j < total_p) {
          int b0 = 0, b1 = nrows - 1;                    // the last row whose pieces start at or before j (empty rows share a start)
          while (b0 < b1) { const int mid = (b0 + b1 + 1) >> 1; if (s_p[mid] <= j) b0 = mid; else b1 = mid - 1; }
          const int q = j - s_p[b0];
          const int64_t st = s_start[b0];
          if (st + s_n[b0] > a.capacity) { over = true; }
          else {
            cnt[u] = min(8, s_n[b0] - 8 * q);
            const S *src = reinterpret_cast<const S *>(a.ids) + (row0 + b0) * (int64_t)a.ld + 8 * q;
            lo[u] = *reinterpret_cast<const uint4 *>(src);
            if (sizeof(S) == 4) hi[u] = *reinterpret_cast<const uint4 *>(src + 4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kPackU; ++u) {
        if (cnt[u] <= 0) continue;
        const int c = cnt[u];
        uint32_t t[8];
        if (sizeof(S) == 4) {
          t[0] = lo[u].x; t[1] = lo[u].y; t[2] = lo[u].z; t[3] = lo[u].w; t[4] = hi[u].x; t[5] = hi[u].y; t[6] = hi[u].z; t[7] = hi[u].w;
        } else {
          t[0] = lo[u].x & 0xFFFFu; t[1] = lo[u].x >> 16; t[2] = lo[u].y & 0xFFFFu; t[3] = lo[u].y >> 16;
          t[4] = lo[u].z & 0xFFFFu; t[5] = lo[u].z >> 16; t[6] = lo[u].w & 0xFFFFu; t[7] = lo[u].w >> 16;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = k < c ? t[k] : 0u;
        if (sizeof(S) == 4) wide |= t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7];
        E *dst = reinterpret_cast<E *>(a.packed) + tbase + 8 * (int64_t)(j0 + u * 256);
        if (sizeof(E) == 2) {
          *reinterpret_cast<uint4 *>(dst) = make_uint4((t[0] & 0xFFFFu) | (t[1] << 16), (t[2] & 0xFFFFu) | (t[3] << 16),
                                                       (t[4] & 0xFFFFu) | (t[5] << 16), (t[6] & 0xFFFFu) | (t[7] << 16));
        } else if (sizeof(E) == 4) {
          *reinterpret_cast<uint4 *>(dst) = make_uint4(t[0], t[1], t[2], t[3]);
          *reinterpret_cast<uint4 *>(dst + 4) = make_uint4(t[4], t[5], t[6], t[7]);
        } else {
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            const int64_t a0 = sizeof(S) == 4 ? (int64_t)(int32_t)t[k] : (int64_t)t[k], a1 = sizeof(S) == 4 ? (int64_t)(int32_t)t[k + 1] : (int64_t)t[k + 1];
            *reinterpret_cast<longlong2 *>(dst + k) = longlong2{a0, a1};
          }
        }
      }
    }
    if (over) atomicOr(a.status, 2);
  } else {
    __syncthreads();
    const int tpr = 1 << a.tpr_shift, sub = tid & (tpr - 1), rpp = 256 >> a.tpr_shift;
    for (int p = tid >> a.tpr_shift; p < kPackTile; p += rpp) {
      const int64_t rr = (int64_t)tile * kPackTile + p;
      if (rr >= a.rows) break;
      const int nn = s_n[p];
      const int64_t st = s_start[p];
      if (st + nn > a.capacity) { over = true; continue; }
      wide |= copy_row_pieces<S, E>(reinterpret_cast<const S *>(a.ids) + rr * (int64_t)a.ld, reinterpret_cast<E *>(a.packed) + st, nn, sub, tpr);
    }
    if (over && sub == 0) atomicOr(a.status, 2);
  }
  if (sizeof(E) == 2 && sizeof(S) == 4 && (wide & 0xFFFF0000u)) atomicOr(a.status, 1);
  __syncthreads();      // the tile's LDS tables are free again
  }
}

// Never reads beyond what the row's owner wrote: a row whose ids would end past its segment (segment_stride: a rank
// whose rows did not fit the caller-given capacity skipped them, gtok_pack_rows status bit 1, while the gathered lengths
// still carry them) or past the buffer (`capacity` elements, 0 = unknown) comes out as all pad and raises status bit 1.
// O: the slab's id type - int32_t (the documented slab) or uint16_t (gtok_unpack_rows_u16: the 16-bit slab GTOK_SENT_U16 writes,
// half the bytes of the re-padding pass that ends every compact all-gather)
template <typename E, typename O>
__global__ void __launch_bounds__(256) unpack_rows_kernel(const RowsArgs a) {
  const int tpr = 1 << a.tpr_shift, sub = (int)threadIdx.x & (tpr - 1);
  const int rpb = 256 >> a.tpr_shift;
  // (a workgroup walks row groups with the grid's stride: at one group of 8 rows per workgroup, 4 M rows were half a million
  // workgroups of 2.8 KB each and the pass ran at 2.4 TB/s of in + out)
  for (int64_t r = (int64_t)blockIdx.x * rpb + ((int)threadIdx.x >> a.tpr_shift); r < a.rows; r += (int64_t)gridDim.x * rpb) {
  int n = a.len[r];
  n = n < 0 ? 0 : (n > a.ld ? a.ld : n);
  const int64_t start = packed_start(a.row_ptr, r, a.segment_rows, a.segment_stride, a.ld, a.at);
  bool fits = start >= 0 && (a.capacity <= 0 || start + n <= a.capacity);
  if (a.row_ptr && a.segment_rows > 0) {
    const int64_t seg = r / a.segment_rows;
    fits = fits && (start - seg * a.segment_stride) + n <= a.segment_stride;
  }
  if (!fits) {
    if (sub == 0 && n > 0 && a.status) atomicOr(a.status, 2);
    n = 0;
  }
  O *__restrict__ row = reinterpret_cast<O *>(a.out_ids) + r * (int64_t)a.ld;
  const E *__restrict__ src = reinterpret_cast<const E *>(a.packed) + (fits ? start : 0);
  const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
  const int pad = a.pad_id;
  for (int i = sub * 8; i < a.ld; i += tpr * 8) {
    if (vec && i + 8 <= a.ld) {
      int32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = pad;
      if (i + 8 <= n) {
        if (sizeof(E) == 2) {
          const uint4 p = *reinterpret_cast<const uint4 *>(src + i);
          v[0] = (int)(p.x & 0xFFFFu); v[1] = (int)(p.x >> 16); v[2] = (int)(p.y & 0xFFFFu); v[3] = (int)(p.y >> 16);
          v[4] = (int)(p.z & 0xFFFFu); v[5] = (int)(p.z >> 16); v[6] = (int)(p.w & 0xFFFFu); v[7] = (int)(p.w >> 16);
        } else {
          const int4 lo = *reinterpret_cast<const int4 *>(src + i), hi = *reinterpret_cast<const int4 *>(src + i + 4);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
      } else if (i < n) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k < n ? (int32_t)src[i + k] : pad;
      }
      if (sizeof(O) == 4) {
        *reinterpret_cast<int4 *>(row + i) = make_int4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<int4 *>(row + i + 4) = make_int4(v[4], v[5], v[6], v[7]);
      } else {
        auto pk = [](int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); };
        *reinterpret_cast<uint4 *>(row + i) = make_uint4(pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7]));
      }
    } else {
      for (int k = i; k < a.ld && k < i + 8; ++k) row[k] = (O)(k < n ? (int32_t)src[k] : pad);
    }
  }
  }
}

// gtok_collate over the packed form: one wave per batch row
template <typename E>
__global__ void __launch_bounds__(256) collate_packed_kernel(const void *__restrict__ packed, const int64_t *__restrict__ row_ptr,
                                                             const int32_t *__restrict__ len, int ld,
                                                             const int64_t *__restrict__ index, int batch, int pad_id,
                                                             int64_t *__restrict__ out_x, uint8_t *__restrict__ out_attn,
                                                             int out_ld) {
  const int lane = lane_id();
  const int b = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  if (b >= batch) return;
  const int64_t src = index[b];
  int n = len[src];
  n = n < 0 ? 0 : (n > ld ? ld : n);
  const E *__restrict__ row = reinterpret_cast<const E *>(packed) + (row_ptr ? row_ptr[src] : src * (int64_t)ld);
  for (int i = lane; i < out_ld; i += kWave) {
    const bool in = i < n;
    out_x[(int64_t)b * out_ld + i] = in ? (int64_t)row[i] : (int64_t)pad_id;
    out_attn[(int64_t)b * out_ld + i] = in ? 1 : 0;
  }
}

// ---- a whole epoch's batches in one launch (gtok_collate_epoch): rows order[0 .. n) cut into batches of batch_size, batch b
// collated to its own width L_b (its longest row) at element batch_off[b] of one arena.  The per-batch route costs a launch,
// two allocations and a host-side maximum per 128 rows - ~25 us of Python and runtime for 13 us of work per ZINC batch.
__global__ void __launch_bounds__(256) collate_plan_kernel(const int32_t *__restrict__ len, int ld, const int64_t *__restrict__ order, int64_t n,
                                                           int batch_size, int64_t nb, int32_t *__restrict__ batch_lmax) {
  const int lane = lane_id();
  const int64_t b = (int64_t)blockIdx.x * 4 + wave_id();
  if (b >= nb) return;
  const int64_t r0 = b * batch_size, r1 = min(n, r0 + batch_size);
  int m = 0;
  for (int64_t r = r0 + lane; r < r1; r += kWave) { int v = len[order[r]]; v = v < 0 ? 0 : (v > ld ? ld : v); m = max(m, v); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
  if (lane == 0) batch_lmax[b] = m;
}

// batch_off[b] = sum over earlier batches of rows x width, batch_off[nb] = the arena's size in elements; one workgroup
__global__ void __launch_bounds__(1024) collate_offsets_kernel(const int32_t *__restrict__ batch_lmax, int64_t n, int batch_size, int64_t nb,
                                                              int64_t *__restrict__ batch_off) {
  __shared__ int64_t s_w[16];
  const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int64_t chunk = (nb + 1023) / 1024, lo = min(nb, tid * chunk), hi = min(nb, lo + chunk);
  auto size_of = [&](int64_t b) -> int64_t { return (min(n, (b + 1) * (int64_t)batch_size) - b * (int64_t)batch_size) * batch_lmax[b]; };
  int64_t s = 0;
  for (int64_t b = lo; b < hi; ++b) s += size_of(b);
  int64_t inc = s;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) { const int64_t up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  if (lane == kWave - 1) s_w[w] = inc;
  __syncthreads();
  int64_t run = inc - s;
  for (int k = 0; k < w; ++k) run += s_w[k];
  for (int64_t b = lo; b < hi; ++b) { batch_off[b] = run; run += size_of(b); }
  if (tid == 1023) batch_off[nb] = run;
}

template <typename E>
__global__ void __launch_bounds__(256) collate_epoch_kernel(const void *__restrict__ packed, const int64_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ len, int ld, const int64_t *__restrict__ order, int64_t n,
                                                            int batch_size, int pad_id, const int32_t *__restrict__ batch_lmax,
                                                            const int64_t *__restrict__ batch_off, int64_t *__restrict__ out_x,
                                                            uint8_t *__restrict__ out_attn, int64_t arena_elems) {
  const int lane = lane_id();
  const int64_t r = (int64_t)blockIdx.x * (int)(blockDim.x >> 6) + wave_id();
  if (r >= n) return;
  const int64_t b = r / batch_size;
  const int L = batch_lmax[b];
  const int64_t at = batch_off[b] + (r - b * batch_size) * (int64_t)L;
  if (at < 0 || at + L > arena_elems) return;                   // (an arena smaller than batch_off[nb]: nothing is written past it)
  const int64_t src = order[r];
  int nn = len[src];
  nn = nn < 0 ? 0 : (nn > ld ? ld : nn);
  const E *__restrict__ row = reinterpret_cast<const E *>(packed) + (row_ptr ? row_ptr[src] : src * (int64_t)ld);
  for (int i = lane; i < L; i += kWave) {
    const bool in = i < nn;
    out_x[at + i] = in ? (int64_t)row[i] : (int64_t)pad_id;
    out_attn[at + i] = in ? 1 : 0;
  }
}

// ---- one batch whose row list arrives from the HOST (gtok_collate_batch): the stock DataLoader hands a Python list of 128
// indices to the dataset per batch; uploading it was a synchronous 1 KB H2D copy and the labels a gather launch of their own.
// Here the indices travel in the kernel's arguments (<= 512 per launch) and the labels are gathered by the same kernel.
constexpr int kBatchArgIdx = 512;
struct BatchArgs {
  const void *packed; const int64_t *row_ptr; const int32_t *len; int64_t *out_x; uint8_t *out_attn;
  const uint8_t *y; uint8_t *out_y;
  int ld, batch, first, pad_id, out_ld, y_bytes;
  int32_t index[kBatchArgIdx];
};

template <typename E>
__global__ void __launch_bounds__(256) collate_batch_kernel(const BatchArgs a) {
  const int lane = lane_id();
  const int k = (int)blockIdx.x * 4 + wave_id();          // row of this launch's chunk
  if (k >= a.batch) return;
  const int64_t src = a.index[k];
  const int b = a.first + k;                              // row of the whole batch
  int n = a.len[src];
  n = n < 0 ? 0 : (n > a.ld ? a.ld : n);
  const E *__restrict__ row = reinterpret_cast<const E *>(a.packed) + (a.row_ptr ? a.row_ptr[src] : src * (int64_t)a.ld);
  for (int i = lane; i < a.out_ld; i += kWave) {
    const bool in = i < n;
    a.out_x[(int64_t)b * a.out_ld + i] = in ? (int64_t)row[i] : (int64_t)a.pad_id;
    a.out_attn[(int64_t)b * a.out_ld + i] = in ? 1 : 0;
  }
  if (a.y && lane < a.y_bytes) a.out_y[(int64_t)b * a.y_bytes + lane] = a.y[src * (int64_t)a.y_bytes + lane];
}

// ---- id rows -> text (the strings ZINCTokenizationDataset.__getitem__ hands to the trainer, zinc_dataset_indexbase.py:143-227,
// rendered for a whole split at once).  One wave per row; lane = token: the row's text is the table strings of its first
// take[r] ids joined by single spaces, then the row's suffix bytes verbatim.
struct TextArgsR {
  const int32_t *ids; int ld; const int32_t *take; int64_t rows;
  const uint8_t *tab_bytes; const int32_t *tab_ptr; int num_strings;
  const uint8_t *suf_bytes; const int64_t *suf_ptr;
  const int64_t *text_ptr; uint8_t *out; int64_t *text_len;
};

__global__ void __launch_bounds__(256) ids_to_text_kernel(const TextArgsR a) {
  const int lane = lane_id();
  const int64_t r = (int64_t)blockIdx.x * (int)(blockDim.x >> 6) + wave_id();
  if (r >= a.rows) return;
  int n = a.take[r];
  n = n < 0 ? 0 : (n > a.ld ? a.ld : n);
  const int32_t *__restrict__ row = a.ids + r * (int64_t)a.ld;
  const int64_t s0 = a.suf_ptr ? a.suf_ptr[r] : 0, s1 = a.suf_ptr ? a.suf_ptr[r + 1] : 0;
  uint8_t *__restrict__ dst = a.out ? a.out + a.text_ptr[r] : nullptr;
  int64_t run = 0;                                   // bytes of the text before this chunk of 64 tokens
  for (int c0 = 0; c0 < n; c0 += kWave) {
    const int i = c0 + lane;
    int len = 0, off = 0;
    if (i < n) {
      const int t = row[i];
      if (t >= 0 && t < a.num_strings) { off = a.tab_ptr[t]; len = a.tab_ptr[t + 1] - off; }
    }
    const int cost = i < n ? len + (i > 0 ? 1 : 0) : 0;   // a space in front of every token but the first
    int inc = cost;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    if (dst && i < n) {
      uint8_t *p = dst + run + (inc - cost);
      if (i > 0) *p++ = (uint8_t)' ';
      const uint8_t *q = a.tab_bytes + off;
      for (int b = 0; b < len; ++b) p[b] = q[b];
    }
    run += __shfl(inc, kWave - 1);
  }
  if (dst) {
    for (int64_t b = lane; b < s1 - s0; b += kWave) dst[run + b] = a.suf_bytes[s0 + b];
  } else if (lane == 0) {
    a.text_len[r] = run + (s1 - s0);
  }
}

// ---- the tail of a ZINC text (zinc_dataset_indexbase.py:186-195: `<p> val_X_XX <eos>`, :217-221: a text of more than
// max_len tokens keeps max_len - 1 of them and `<eos>`).  The serialiser's rows end with `<p>`; what follows differs per
// molecule, so it is rendered here, thread per row, instead of one Python format call per molecule on the host.
// `f"val_{label:.2f}".replace('.', '_').replace('-', 'neg')` (:192) on the float32 label: "%.2f" of the EXACT value, ties to
// even - 24 bits of mantissa times 100 fit 31 bits, so below 2^24 the hundredths are one shift with a remainder test, and
// from 2^23 on the value is an integer of up to 128 bits, printed in full as Python does.
__device__ int zinc_label_token(float v, uint8_t *buf) {            // buf: >= 49 bytes; returns the length
  const uint32_t b = __float_as_uint(v);
  const bool neg = (b >> 31) != 0u;
  const uint32_t ex = (b >> 23) & 255u, mant = b & 0x7FFFFFu;
  int n = 0;
  buf[n++] = 'v'; buf[n++] = 'a'; buf[n++] = 'l'; buf[n++] = '_';
  if (ex == 255u && mant) { buf[n++] = 'n'; buf[n++] = 'a'; buf[n++] = 'n'; return n; }   // (Python prints nan without a sign)
  if (neg) { buf[n++] = 'n'; buf[n++] = 'e'; buf[n++] = 'g'; }     // the sign bit: -0.001 is val_neg0_00
  if (ex == 255u) { buf[n++] = 'i'; buf[n++] = 'n'; buf[n++] = 'f'; return n; }
  const uint32_t m = ex ? (mant | 0x800000u) : mant;
  const int e = (ex ? (int)ex : 1) - 150;                           // value = m * 2^e
  uint32_t limb[4] = {0u, 0u, 0u, 0u};                              // integer part, least significant first
  uint32_t frac = 0u;
  if (e >= 0) {
    const int w = e >> 5, sh = e & 31;                              // e <= 104: w <= 3, and at w == 3 the shift is <= 8
    limb[w] = m << sh;
    if (sh && w < 3) limb[w + 1] = m >> (32 - sh);
  } else {
    const int sh = -e;
    uint32_t h = 0u;
    if (sh < 64) {
      const uint64_t q = (uint64_t)m * 100u;
      h = (uint32_t)(q >> sh);
      const uint64_t rem = q & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
      if (rem > half || (rem == half && (h & 1u))) ++h;
    }
    limb[0] = h / 100u; frac = h % 100u;
  }
  uint8_t dig[40];
  int nd = 0;
  do {
    uint32_t r = 0u;
    for (int k = 3; k >= 0; --k) { const uint64_t cur = ((uint64_t)r << 32) | limb[k]; limb[k] = (uint32_t)(cur / 10u); r = (uint32_t)(cur % 10u); }
    dig[nd++] = (uint8_t)('0' + r);
  } while (limb[0] | limb[1] | limb[2] | limb[3]);
  for (int k = nd - 1; k >= 0; --k) buf[n++] = dig[k];
  buf[n++] = '_'; buf[n++] = (uint8_t)('0' + frac / 10u); buf[n++] = (uint8_t)('0' + frac % 10u);
  return n;
}

struct TailArgs {
  const float *y; const int32_t *len; int64_t rows; int max_len;
  int32_t *take; const int64_t *suf_ptr; uint8_t *out; int64_t *suf_len;
};

__global__ void __launch_bounds__(256) zinc_text_tails_kernel(const TailArgs a) {
  const int64_t r = (int64_t)blockIdx.x * 256 + (int)threadIdx.x;
  if (r >= a.rows) return;
  const int n = max(a.len[r], 0);
  const bool cut = (int64_t)n + 2 > (int64_t)a.max_len;             // tokens = the row's ids + [label, <eos>]
  const int take = cut ? a.max_len - 1 : n;
  uint8_t buf[64];
  int k = 0;
  if (take > 0) buf[k++] = ' ';
  if (!cut) { k += zinc_label_token(a.y[r], buf + k); buf[k++] = ' '; }
  buf[k++] = '<'; buf[k++] = 'e'; buf[k++] = 'o'; buf[k++] = 's'; buf[k++] = '>';
  if (a.out) {
    uint8_t *__restrict__ p = a.out + a.suf_ptr[r];
    for (int j = 0; j < k; ++j) p[j] = buf[j];
  } else {
    a.take[r] = take;
    a.suf_len[r] = k;
  }
}

static int tpr_shift_for(int ld) {
  const int pieces = (ld + 7) / 8;
  int s = 0;
  while ((1 << s) < pieces && s < 6) ++s;
  return s;
}

}  // namespace gtok

using namespace gtok;

extern "C" int gtok_row_offsets(const int32_t *len, int64_t num_rows, int32_t ld, int32_t align, int64_t *row_ptr,
                                void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (num_rows < 0 || ld <= 0 || align < 1 || (align & (align - 1)) || !row_ptr) return GTOK_E_INVAL;
  if (num_rows > 0 && !len) return GTOK_E_INVAL;
  const int64_t tiles = (num_rows + kScanTile - 1) / kScanTile;
  if (tiles > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  hipStream_t s = (hipStream_t)stream;
  if (tiles > 0)
    hipLaunchKernelGGL(row_tile_sums_kernel, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, len, num_rows, ld, align - 1, row_ptr);
  hipLaunchKernelGGL(row_tile_scan_kernel, dim3(1), dim3(kScanBlock), 0, s, num_rows, row_ptr);
  if (tiles > 0)
    hipLaunchKernelGGL(row_tile_fill_kernel, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, len, num_rows, ld, align - 1, row_ptr);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

#ifndef GTOK_UNPACK_WG_PER_CU
#define GTOK_UNPACK_WG_PER_CU 32
#endif
constexpr int kUnpackWgPerCu = GTOK_UNPACK_WG_PER_CU;   // workgroups per CU of the re-padding pass; beyond that they stride over the rows (tuning knob)

static int rows_launch(bool pack, const void *ids, int src_bytes, void *out_ids, int out_bytes, int32_t ld, const int32_t *len, int64_t num_rows,
                       const int64_t *row_ptr, int32_t segment_rows, int64_t segment_stride, int32_t elem_bytes,
                       void *packed, int64_t capacity, int32_t pad_id, int32_t *status, void *stream, int at = 0) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  const bool eb_ok = elem_bytes == 2 || elem_bytes == 4 || (pack && src_bytes == 2 && elem_bytes == 8);
  if (num_rows < 0 || ld <= 0 || !eb_ok || segment_stride < 0 || segment_rows < 0) return GTOK_E_INVAL;
  if (num_rows == 0) return GTOK_OK;
  if (!len || !packed || (pack ? (!ids || !row_ptr) : !out_ids) || (pack && !status) || capacity < 0) return GTOK_E_INVAL;
  if (!pack && !row_ptr && segment_rows > 0) return GTOK_E_INVAL;   // the strided form has no segments
  RowsArgs a;
  a.ids = ids; a.out_ids = out_ids; a.len = len; a.row_ptr = row_ptr; a.packed = packed; a.status = status;
  a.rows = num_rows; a.ld = ld; a.pad_id = pad_id; a.segment_rows = segment_rows; a.segment_stride = segment_stride; a.capacity = capacity;
  a.tpr_shift = tpr_shift_for(ld);
  if (!pack) {
    // the re-padding pass: half as many threads per row as it has 16-byte pieces, two pieces each (ZINC-full x 16 epochs into a
    // 16-bit slab: 0.52 ms against 0.67 with a thread per piece - 22 pieces on 32 threads left a third of the lanes idle)
    int less = 1;
    if (const char *cs = std::getenv("GTOK_UNPACK_TPR_LESS")) less = std::atoi(cs);     // tuning knob
    while (less-- > 0 && a.tpr_shift > 0) --a.tpr_shift;
  }
  a.at = at;
  const int rpb = 256 >> a.tpr_shift;
  const int64_t nb = (num_rows + rpb - 1) / rpb;
  if (nb > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  hipStream_t s = (hipStream_t)stream;
  if (pack && src_bytes == 2) {
    if (elem_bytes == 2) hipLaunchKernelGGL((pack_rows_kernel<uint16_t, uint16_t>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else if (elem_bytes == 4) hipLaunchKernelGGL((pack_rows_kernel<uint16_t, int32_t>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pack_rows_kernel<uint16_t, int64_t>), dim3((unsigned)nb), dim3(256), 0, s, a);
  } else if (pack) {
    if (elem_bytes == 2) hipLaunchKernelGGL((pack_rows_kernel<int32_t, uint16_t>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pack_rows_kernel<int32_t, int32_t>), dim3((unsigned)nb), dim3(256), 0, s, a);
  } else {
    int per_cu = kUnpackWgPerCu;
    if (const char *cs = std::getenv("GTOK_UNPACK_WG_PER_CU")) { const int c = std::atoi(cs); if (c >= 0) per_cu = c; }   // tuning knob: 0 = one row group per workgroup
    const int64_t cap_wg = per_cu > 0 ? (int64_t)device_cu_count(device_scope.dev) * per_cu : nb;
    const unsigned nu = (unsigned)(nb < cap_wg ? nb : cap_wg);
    if (out_bytes == 2) {
      if (elem_bytes == 2) hipLaunchKernelGGL((unpack_rows_kernel<uint16_t, uint16_t>), dim3(nu), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((unpack_rows_kernel<int32_t, uint16_t>), dim3(nu), dim3(256), 0, s, a);
    } else {
      if (elem_bytes == 2) hipLaunchKernelGGL((unpack_rows_kernel<uint16_t, int32_t>), dim3(nu), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((unpack_rows_kernel<int32_t, int32_t>), dim3(nu), dim3(256), 0, s, a);
    }
  }
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_pack_rows_scan(const void *ids, int32_t src_bytes, int32_t ld, const int32_t *len, int64_t num_rows, int32_t align,
                                   int32_t elem_bytes, void *packed, int64_t capacity, int64_t *row_ptr, int32_t *status, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  const bool eb_ok = elem_bytes == 2 || elem_bytes == 4 || (src_bytes == 2 && elem_bytes == 8);
  if (num_rows < 0 || ld <= 0 || (src_bytes != 2 && src_bytes != 4) || !eb_ok || align < 1 || (align & (align - 1)) || capacity < 0 || !row_ptr)
    return GTOK_E_INVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!status || (num_rows > 0 && (!ids || !len || !packed))) return GTOK_E_INVAL;
  const int ncu = device_cu_count(device_scope.dev);
  int tile_rows = 256;      // (smaller tiles were measured slower at every size: more tickets, more look-back)
  if (const char *cs = std::getenv("GTOK_PACK_TILE")) { const int c = std::atoi(cs); if (c == 64 || c == 128 || c == 256) tile_rows = c; }   // tuning knob
  const int64_t tiles = (num_rows + tile_rows - 1) / tile_rows;
  if (tiles > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  hipLaunchKernelGGL(pack_scan_init_kernel, dim3((unsigned)(tiles > 0 ? (tiles + 255) / 256 : 1)), dim3(256), 0, s, row_ptr, num_rows, (int)tiles, tile_rows, status);
  if (num_rows == 0) return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
  QueueSlot slot = take_queue_slot(device_scope.dev, s);       // the tile ticket: a counter of this launch's own (zero between launches)
  if (!slot.counters) return slot.graph_pool_empty ? GTOK_E_GRAPH_SLOTS : GTOK_E_LAUNCH;
  PackScanArgs a;
  a.ids = ids; a.len = len; a.row_ptr = row_ptr; a.packed = packed; a.status = status; a.ticket = slot.counters;
  a.rows = num_rows; a.capacity = capacity; a.ld = ld; a.align_mask = align - 1; a.tiles = (int)tiles;
  // (general path) threads per row: enough for half the widest row's 8-id pieces in one pass (rows are ~half as long as the slab is wide)
  int sh = tpr_shift_for(ld);
  if (sh > 0) --sh;
  a.tpr_shift = sh;
  a.fast = align == 8 && ((int64_t)ld * src_bytes) % 16 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(ids) & 15u) == 0 &&
           (reinterpret_cast<uintptr_t>(packed) & 15u) == 0;
  a.chunk = (int)((tiles + 8ll * ncu - 1) / (8ll * ncu));
  if (a.chunk < 1) a.chunk = 1;
  if (a.chunk > kPackChunkMax) a.chunk = kPackChunkMax;
  if (const char *cs = std::getenv("GTOK_PACK_CHUNK")) { const int c = std::atoi(cs); if (c >= 1 && c <= kPackChunkMax) a.chunk = c; }   // tuning knob
  const dim3 grid((unsigned)((tiles + a.chunk - 1) / a.chunk)), block(256);
#define GTOK_PACK_SCAN(S_, E_)                                                                              \
  do {                                                                                                      \
    if (tile_rows == 256) hipLaunchKernelGGL((pack_scan_kernel<S_, E_, 256>), grid, block, 0, s, a);        \
    else if (tile_rows == 128) hipLaunchKernelGGL((pack_scan_kernel<S_, E_, 128>), grid, block, 0, s, a);   \
    else hipLaunchKernelGGL((pack_scan_kernel<S_, E_, 64>), grid, block, 0, s, a);                          \
  } while (0)
  if (src_bytes == 2) {
    if (elem_bytes == 2) GTOK_PACK_SCAN(uint16_t, uint16_t);
    else if (elem_bytes == 4) GTOK_PACK_SCAN(uint16_t, int32_t);
    else GTOK_PACK_SCAN(uint16_t, int64_t);
  } else {
    if (elem_bytes == 2) GTOK_PACK_SCAN(int32_t, uint16_t);
    else GTOK_PACK_SCAN(int32_t, int32_t);
  }
#undef GTOK_PACK_SCAN
  mark_queue_slot(slot, s);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_pack_rows(const int32_t *ids, int32_t ld, const int32_t *len, int64_t num_rows, const int64_t *row_ptr,
                              int32_t elem_bytes, void *packed, int64_t capacity, int32_t *status, void *stream) {
  return rows_launch(true, ids, 4, nullptr, 4, ld, len, num_rows, row_ptr, 0, 0, elem_bytes, packed, capacity, 0, status, stream);
}

extern "C" int gtok_pack_rows_u16(const uint16_t *ids16, int32_t ld, const int32_t *len, int64_t num_rows, const int64_t *row_ptr,
                                  int32_t elem_bytes, void *packed, int64_t capacity, int32_t *status, void *stream) {
  return rows_launch(true, ids16, 2, nullptr, 4, ld, len, num_rows, row_ptr, 0, 0, elem_bytes, packed, capacity, 0, status, stream);
}

extern "C" int gtok_unpack_rows(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                                int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int32_t pad_id,
                                int32_t *out_ids, int32_t ld, void *stream) {
  return rows_launch(false, nullptr, 4, out_ids, 4, ld, len, num_rows, row_ptr, segment_rows, segment_stride, elem_bytes,
                     const_cast<void *>(packed), 0, pad_id, nullptr, stream);
}

extern "C" int gtok_unpack_rows_checked(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                                        int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int64_t packed_elems,
                                        int32_t pad_id, int32_t *out_ids, int32_t ld, int32_t *status, void *stream) {
  return rows_launch(false, nullptr, 4, out_ids, 4, ld, len, num_rows, row_ptr, segment_rows, segment_stride, elem_bytes,
                     const_cast<void *>(packed), packed_elems, pad_id, status, stream);
}

extern "C" int gtok_unpack_rows_u16(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                                    int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int64_t packed_elems,
                                    int32_t pad_id, uint16_t *out_ids16, int32_t ld, int32_t *status, void *stream) {
  if (pad_id < 0 || pad_id > 65535) return GTOK_E_INVAL;
  return rows_launch(false, nullptr, 4, out_ids16, 2, ld, len, num_rows, row_ptr, segment_rows, segment_stride, elem_bytes,
                     const_cast<void *>(packed), packed_elems, pad_id, status, stream);
}

extern "C" int gtok_unpack_rows_at(const void *packed, int32_t elem_bytes, const int64_t *row_start, const int32_t *len, int64_t num_rows,
                                   int32_t segment_rows, int64_t segment_stride, int64_t packed_elems, int32_t pad_id, void *out_ids,
                                   int32_t out_bytes, int32_t ld, int32_t *status, void *stream) {
  if ((out_bytes != 2 && out_bytes != 4) || !row_start || (out_bytes == 2 && (pad_id < 0 || pad_id > 65535))) return GTOK_E_INVAL;
  return rows_launch(false, nullptr, 4, out_ids, out_bytes, ld, len, num_rows, row_start, segment_rows, segment_stride, elem_bytes,
                     const_cast<void *>(packed), packed_elems, pad_id, status, stream, 1);
}

extern "C" int gtok_collate_packed(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                                   int32_t ld, const int64_t *index, int32_t batch, int32_t pad_id, int64_t *out_x,
                                   uint8_t *out_attn, int32_t out_ld, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!packed || !len || !index || ld <= 0 || batch < 0 || out_ld < 0 || (elem_bytes != 2 && elem_bytes != 4))
    return GTOK_E_INVAL;      // (row_ptr == NULL: the strided form)
  if (batch == 0 || out_ld == 0) return GTOK_OK;
  if (!out_x || !out_attn) return GTOK_E_INVAL;
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 2)
    hipLaunchKernelGGL(collate_packed_kernel<uint16_t>, dim3((batch + 3) / 4), dim3(256), 0, s, packed, row_ptr, len, ld, index,
                       batch, pad_id, out_x, out_attn, out_ld);
  else
    hipLaunchKernelGGL(collate_packed_kernel<int32_t>, dim3((batch + 3) / 4), dim3(256), 0, s, packed, row_ptr, len, ld, index,
                       batch, pad_id, out_x, out_attn, out_ld);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_collate_batch(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len, int32_t ld,
                                  const int64_t *host_index, int32_t batch, int64_t num_rows, int32_t pad_id, int64_t *out_x, uint8_t *out_attn,
                                  int32_t out_ld, const void *y, int32_t y_bytes, void *out_y, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (batch < 0 || ld <= 0 || out_ld < 0 || num_rows < 0 || (elem_bytes != 2 && elem_bytes != 4) || (y && (y_bytes < 1 || y_bytes > 8 || !out_y)))
    return GTOK_E_INVAL;
  if (batch == 0) return GTOK_OK;
  if (!packed || !len || !host_index || (out_ld > 0 && (!out_x || !out_attn))) return GTOK_E_INVAL;
  for (int i = 0; i < batch; ++i)
    if (host_index[i] < 0 || host_index[i] >= num_rows) return GTOK_E_INVAL;       // the list is on the host: checked before anything is launched
  hipStream_t s = (hipStream_t)stream;
  BatchArgs a;
  a.packed = packed; a.row_ptr = row_ptr; a.len = len; a.out_x = out_x; a.out_attn = out_attn;
  a.y = reinterpret_cast<const uint8_t *>(y); a.out_y = reinterpret_cast<uint8_t *>(out_y);
  a.ld = ld; a.pad_id = pad_id; a.out_ld = out_ld; a.y_bytes = y ? y_bytes : 0;
  for (int first = 0; first < batch; first += kBatchArgIdx) {
    const int cnt = batch - first < kBatchArgIdx ? batch - first : kBatchArgIdx;
    a.batch = cnt; a.first = first;
    for (int i = 0; i < cnt; ++i) a.index[i] = (int32_t)host_index[first + i];
    if (elem_bytes == 2) hipLaunchKernelGGL(collate_batch_kernel<uint16_t>, dim3((cnt + 3) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(collate_batch_kernel<int32_t>, dim3((cnt + 3) / 4), dim3(256), 0, s, a);
  }
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_collate_epoch_plan(const int32_t *len, int32_t ld, const int64_t *order, int64_t n, int32_t batch_size,
                                       int32_t *batch_lmax, int64_t *batch_off, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (n < 0 || ld <= 0 || batch_size <= 0 || !batch_off) return GTOK_E_INVAL;
  const int64_t nb = (n + batch_size - 1) / batch_size;
  if (nb > 0 && (!len || !order || !batch_lmax)) return GTOK_E_INVAL;
  if ((nb + 3) / 4 > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  hipStream_t s = (hipStream_t)stream;
  if (nb > 0) hipLaunchKernelGGL(collate_plan_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, len, ld, order, n, batch_size, nb, batch_lmax);
  hipLaunchKernelGGL(collate_offsets_kernel, dim3(1), dim3(1024), 0, s, batch_lmax, n, batch_size, nb, batch_off);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_collate_epoch(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len, int32_t ld,
                                  const int64_t *order, int64_t n, int32_t batch_size, int32_t pad_id, const int32_t *batch_lmax,
                                  const int64_t *batch_off, int64_t *out_x, uint8_t *out_attn, int64_t arena_elems, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (n < 0 || ld <= 0 || batch_size <= 0 || arena_elems < 0 || (elem_bytes != 2 && elem_bytes != 4)) return GTOK_E_INVAL;
  if (n == 0) return GTOK_OK;
  if (!packed || !len || !order || !batch_lmax || !batch_off || !out_x || !out_attn) return GTOK_E_INVAL;
  if ((n + 3) / 4 > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
  if (elem_bytes == 2)
    hipLaunchKernelGGL(collate_epoch_kernel<uint16_t>, grid, block, 0, s, packed, row_ptr, len, ld, order, n, batch_size, pad_id, batch_lmax, batch_off,
                       out_x, out_attn, arena_elems);
  else
    hipLaunchKernelGGL(collate_epoch_kernel<int32_t>, grid, block, 0, s, packed, row_ptr, len, ld, order, n, batch_size, pad_id, batch_lmax, batch_off,
                       out_x, out_attn, arena_elems);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_ids_to_text(const int32_t *ids, int32_t ld, const int32_t *take, int64_t num_rows, const uint8_t *tab_bytes,
                                const int32_t *tab_ptr, int32_t num_strings, const uint8_t *suf_bytes, const int64_t *suf_ptr,
                                const int64_t *text_ptr, uint8_t *out_bytes, int64_t *text_len, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (num_rows < 0 || ld <= 0 || num_strings < 0) return GTOK_E_INVAL;
  if (num_rows == 0) return GTOK_OK;
  if (!ids || !take || !tab_bytes || !tab_ptr || (suf_ptr && !suf_bytes)) return GTOK_E_INVAL;
  if (out_bytes ? !text_ptr : !text_len) return GTOK_E_INVAL;
  const int64_t nb = (num_rows + 3) / 4;
  if (nb > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  TextArgsR a;
  a.ids = ids; a.ld = ld; a.take = take; a.rows = num_rows; a.tab_bytes = tab_bytes; a.tab_ptr = tab_ptr; a.num_strings = num_strings;
  a.suf_bytes = suf_bytes; a.suf_ptr = suf_ptr; a.text_ptr = text_ptr; a.out = out_bytes; a.text_len = text_len;
  hipLaunchKernelGGL(ids_to_text_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" int gtok_zinc_text_tails(const float *y, const int32_t *len, int64_t num_rows, int32_t max_len, int32_t *take,
                                    const int64_t *suf_ptr, uint8_t *suf_bytes, int64_t *suf_len, void *stream) {
  DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (num_rows < 0 || max_len < 1) return GTOK_E_INVAL;
  if (num_rows == 0) return GTOK_OK;
  if (!y || !len || (suf_bytes ? !suf_ptr : (!take || !suf_len))) return GTOK_E_INVAL;
  const int64_t nb = (num_rows + 255) / 256;
  if (nb > 0x7FFFFFFF) return GTOK_E_TOO_LARGE;
  TailArgs a;
  a.y = y; a.len = len; a.rows = num_rows; a.max_len = max_len; a.take = take; a.suf_ptr = suf_ptr; a.out = suf_bytes; a.suf_len = suf_len;
  hipLaunchKernelGGL(zinc_text_tails_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}
