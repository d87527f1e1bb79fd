// gtok_common.hpp — wave-level helpers shared by the gfx950 tokenizer kernels.
//
// Execution model used by every kernel in this directory: ONE 64-lane
// wavefront owns one graph (or one text); the 1..4 waves of a workgroup never
// synchronise with each other, each carves its own slice out of the
// workgroup's dynamic LDS.  Cross-lane hand-offs through LDS are ordered by
// wave_sync() (LDS ops of one wave execute in issue order; the fence pair only
// stops the compiler from moving a lane's load above another lane's store).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace gtok {

constexpr int kWave = 64;

// Workgroups per CU to size a one-round grid for.  These kernels carry ~100 SGPRs, for which gfx950 admits
// floor(800 / (ceil(sgpr/16)*16 + 16)) = 6 waves per SIMD while hipOccupancyMaxActiveBlocksPerMultiprocessor
// answers 8 (MI355X_MICROARCH.md, "Residency and cooperative launch"); a grid sized by the API's answer runs
// a straggler second round (measured: 0.72 ms at 8 vs 0.65 ms at 6).  GTOK_MAX_BLOCKS_PER_CU overrides.
inline int resident_blocks(int api_answer) {
  int cap = 6;
  if (const char *s = std::getenv("GTOK_MAX_BLOCKS_PER_CU")) {
    const int c = std::atoi(s);
    if (c >= 1) cap = c;
  }
  return api_answer < cap ? api_answer : cap;
}

// same limit for kernels launched as one-wave workgroups: 4 SIMDs x 6 waves
inline int resident_waves(int api_answer) {
  int cap = 24;
  if (const char *s = std::getenv("GTOK_MAX_WAVES_PER_CU")) {
    const int c = std::atoi(s);
    if (c >= 1) cap = c;
  }
  return api_answer < cap ? api_answer : cap;
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// wave index inside the workgroup, as a SCALAR: threadIdx.x >> 6 is the same in all 64 lanes but the
// compiler cannot see that, and everything derived from it (graph index, sizes, loop control) would
// otherwise be computed per lane with exec-mask branches
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16-byte vectors for staging and fills.  I32x4 / U8x16 carry only the element alignment of the array they are
// read from (a CSR chunk starts at an arbitrary element; gfx950 global accesses may be unaligned), U8x16a is
// for 16-byte aligned LDS.
struct __attribute__((packed, aligned(4))) I32x4 { int32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) I32x2 { int32_t x, y; };
struct __attribute__((packed, aligned(1))) U8x16 { uint32_t a, b, c, d; };
struct __attribute__((aligned(16))) U8x16a { uint32_t a, b, c, d; };

// ---- dynamic work distribution ------------------------------------------------------------------------
// Work units (graphs, or 64-graph groups) are handed to waves in index order from device-wide ticket counters:
// walk length varies several-fold inside a batch, and a static split leaves waves idle behind the slowest.
// One counter serialises at ~15 ns per draw (measured: 0.48 ms for 32768 draws), so a launch uses kQueues of
// them on separate cache lines.  Every wave's first unit is its own index (no draw); ticket t of queue q is
// unit first_free + t * kQueues + q.  A wave draws from its home queue and moves round-robin to the next when
// one runs dry; the next ticket is drawn before the current unit is processed, which hides the round trip.
// Counter block (host: take_queue_slot): kQueues ticket counters, one retired-groups counter and kQueues
// retired-waves counters, kQueueStride ints apart, all zero between launches - the last wave to retire re-arms it.
constexpr int kQueues = 16, kQueueStride = 32;   // 128 B apart
struct Tickets {
  int *ctr;
  int q, home_q, first_free, limit;
  __device__ __forceinline__ void init(int *counters, int wave_index, int num_waves, int num_units) {
    ctr = counters; q = home_q = wave_index & (kQueues - 1); first_free = num_waves; limit = num_units;
  }
  __device__ __forceinline__ int draw(bool lane0) const {   // the ticket lands in lane 0's register
    int t = 0;
    if (lane0) t = atomicAdd(ctr + q * kQueueStride, 1);
    return t;
  }
  // ticket -> unit; `limit` when every queue is dry.  A dry home queue is not followed by a walk over the other
  // fifteen (fifteen dependent atomics, ~40 us, at the very end of every wave's life): lanes 0..15 read the sixteen
  // counters at once, and the wave moves straight to a queue that still has tickets, or stops.  (A counter only
  // grows, so a stale read can show tickets that are gone - the draw then comes back dry and the wave looks again -
  // but never hides one.)
  __device__ __forceinline__ int settle(int t, bool lane0) {
    int u = first_free + uni(t) * kQueues + q;
    while (u >= limit) {
      const int lane = lane_id();
      int left = 0;
      if (lane < kQueues) {
        const int cap = (limit - first_free - lane + kQueues - 1) / kQueues;   // tickets queue `lane` can hand out
        left = cap - __hip_atomic_load(ctr + lane * kQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const uint64_t live = __ballot(left > 0);
      if (!live) return limit;
      const uint64_t ahead = live & ~((2ull << q) - 1ull);                     // prefer the next queue after q
      q = ahead ? __builtin_ctzll(ahead) : __builtin_ctzll(live);
      u = first_free + uni(draw(lane0)) * kQueues + q;
    }
    return u;
  }
  // After its last draw every wave retires; the last one out re-arms the block for the next launch that gets it.
  // One retired-waves counter would be hit by every wave of the grid (6144 atomics on one address = ~90 us when
  // they bunch up - all of a small launch): the waves retire in kQueues groups (by home queue) on separate lines,
  // and only each group's last wave touches the global counter.
  __device__ __forceinline__ void retire(bool lane0, int lane, int num_waves) const {
    const int home = home_q;
    const int group = (num_waves - home + kQueues - 1) / kQueues;       // waves whose index is home mod kQueues
    int last = 0;
    if (lane0) {
      if (atomicAdd(ctr + (kQueues + 1 + home) * kQueueStride, 1) == group - 1) {
        const int groups = num_waves < kQueues ? num_waves : kQueues;
        last = atomicAdd(ctr + kQueues * kQueueStride, 1) == groups - 1;
      }
    }
    if (uni(last) && lane < 2 * kQueues + 1)
      __hip_atomic_store(ctr + lane * kQueueStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// Ticket counters of the dynamically scheduled kernels: a per-device ring of slots (kQueues counters + the retire
// counters each), zeroed once when the device is first used.  A launch takes the next slot and its last wave re-arms
// it, so a slot is clean again when its launch has drained.  A slot is only handed out again once the launch that
// used it last has FINISHED: every slot carries a HIP event recorded behind its launch, and take_queue_slot waits for
// that event when the ring has wrapped onto a launch still in flight (256+ launches queued without a
// synchronisation) - two live launches never share counters.  Launches issued while the stream is being CAPTURED
// into a hipGraph get a block of their own from a reserved pool (a replay must find the counters it was captured with,
// and event queries are not capturable), which the graph hands back when it is destroyed (release_graph_slot): at most
// kGraphSlots captured launches of the ticket-scheduled kernels in LIVE graphs per device - one more fails with
// GTOK_E_GRAPH_SLOTS.  The first call on a device allocates (not capturable: warm up once before capturing).
struct QueueSlot {
  int *counters = nullptr;
  int index = -1;       // ring index, -1: reserved (captured) slot - nothing to record
  int dev = 0;
  bool graph_pool_empty = false;   // a captured launch found all kGraphSlots reserved slots taken (GTOK_E_GRAPH_SLOTS)
};
constexpr int kSlots = 256, kGraphSlots = 64, kMaxDev = 64, kSlotInts = (2 * kQueues + 1) * kQueueStride;
struct QueueRing {
  std::mutex mu;
  int *mem[kMaxDev] = {};
  hipEvent_t ev[kMaxDev][kSlots] = {};
  bool used[kMaxDev][kSlots] = {};
  unsigned seq[kMaxDev] = {};
  bool graph_used[kMaxDev][kGraphSlots] = {};   // reserved blocks held by live hipGraphs
  bool graph_keep[kMaxDev][kGraphSlots] = {};   // ... whose hand-over to a graph failed: never returned (a graph may still replay with them)
};
inline QueueRing &queue_ring() { static QueueRing r; return r; }

// A reserved block goes back to the pool when the hipGraph that captured its launch is destroyed (and every executable
// graph instantiated from it: they hold their own reference): a HIP user object owned by the graph carries the block's
// (device, index) and hands it back from its destructor - which runs on a runtime thread and calls no HIP function.  The
// block is clean then: the last wave of every launch re-arms its counters - PROVIDED no replay of an executable graph made from
// that graph is still running: synchronise the streams its replays were launched on before destroying a graph (the counters of a
// replay in flight would otherwise be handed to the next capture).
inline void release_graph_slot(void *tag) {
  const uintptr_t t = reinterpret_cast<uintptr_t>(tag) - 1;
  const int dev = (int)(t / kGraphSlots), idx = (int)(t % kGraphSlots);
  if (dev < 0 || dev >= kMaxDev) return;
  QueueRing &r = queue_ring();
  std::lock_guard<std::mutex> lock(r.mu);
  if (!r.graph_keep[dev][idx]) r.graph_used[dev][idx] = false;
}

inline QueueSlot take_queue_slot(int dev, hipStream_t stream) {
  QueueSlot out;
  if (dev < 0 || dev >= kMaxDev) return out;
  QueueRing &r = queue_ring();
  hipEvent_t wait_for = nullptr;
  int graph_idx = -1;                                   // a reserved block taken for a captured launch
  {
    std::lock_guard<std::mutex> lock(r.mu);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (!r.mem[dev]) {
      if (cap != hipStreamCaptureStatusNone) return out;                    // cannot allocate inside a capture
      int *p = nullptr;
      const size_t bytes = (size_t)(kSlots + kGraphSlots) * kSlotInts * sizeof(int);
      if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess) return out;
      if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return out; }
      r.mem[dev] = p;
    }
    out.dev = dev;
    if (cap != hipStreamCaptureStatusNone) {
      int idx = -1;
      for (int i = 0; i < kGraphSlots; ++i)
        if (!r.graph_used[dev][i]) { idx = i; break; }
      if (idx < 0) { out.graph_pool_empty = true; return out; }
      r.graph_used[dev][idx] = true;                    // reserved under the lock; the hand-over to the graph happens outside it
      graph_idx = idx;
      out.counters = r.mem[dev] + (size_t)kSlotInts * (kSlots + idx);
    } else {
    // the ring index is this call's alone until the ring wraps again (kSlots launches later): the wait for the launch
    // that used it last happens OUTSIDE the lock, so launches on other devices and threads are not held up behind it
    const int i = (int)(r.seq[dev]++ % kSlots);
    if (r.used[dev][i]) {
      if (hipEventQuery(r.ev[dev][i]) != hipSuccess) { (void)hipGetLastError(); wait_for = r.ev[dev][i]; }
    } else if (!r.ev[dev][i]) {
      if (hipEventCreateWithFlags(&r.ev[dev][i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return out; }
    }
    out.counters = r.mem[dev] + (size_t)kSlotInts * i;
    out.index = i;
    }
  }
  if (graph_idx >= 0) {
    // tie the block to the capturing graph's lifetime - with the ring's mutex RELEASED: the user object's destructor
    // (release_graph_slot) takes that mutex from a runtime thread, possibly under a lock of the runtime's own that the calls
    // below need as well (an ABBA order with a hipGraphDestroy on another thread).  On any failure the block stays
    // reserved for the life of the process (never handed to a second graph) and a user object that no graph retained is released.
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipGraph_t graph = nullptr;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, nullptr, nullptr) == hipSuccess && graph) {
      hipUserObject_t obj = nullptr;
      void *tag = reinterpret_cast<void *>((uintptr_t)dev * kGraphSlots + graph_idx + 1);
      if (hipUserObjectCreate(&obj, tag, release_graph_slot, 1, hipUserObjectNoDestructorSync) == hipSuccess) {
        if (hipGraphRetainUserObject(graph, obj, 1, hipGraphUserObjectMove) != hipSuccess) {
          (void)hipGetLastError();
          { std::lock_guard<std::mutex> lock(r.mu); r.graph_keep[dev][graph_idx] = true; }   // the captured launch still uses the block
          (void)hipUserObjectRelease(obj, 1);           // (its destructor then leaves the block reserved)
        }
      } else {
        (void)hipGetLastError();
      }
    } else {
      (void)hipGetLastError();
    }
    return out;
  }
  if (wait_for && hipEventSynchronize(wait_for) != hipSuccess) {      // still in flight (or unknown): wait for that launch
    (void)hipGetLastError();
    out.counters = nullptr;
  }
  return out;
}
// behind the launch that uses the slot, on the launch's stream
inline void mark_queue_slot(const QueueSlot &q, hipStream_t stream) {
  if (q.index < 0 || !q.counters) return;
  QueueRing &r = queue_ring();
  std::lock_guard<std::mutex> lock(r.mu);
  r.used[q.dev][q.index] = hipEventRecord(r.ev[q.dev][q.index], stream) == hipSuccess;
}

// The device a launch belongs to is the STREAM's, not the calling thread's current device (a batch on cuda:1 launched
// while cuda:0 is current would otherwise get cuda:0's counters and occupancy): DeviceScope makes it current for the
// duration of the call and restores the caller's device afterwards.  ok() is false when no device could be determined.
struct DeviceScope {
  int dev = -1, prev = -1;
  explicit DeviceScope(hipStream_t stream) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; return; }
    dev = prev;
    if (stream) {
      int sd = -1;
      if (hipStreamGetDevice(stream, &sd) == hipSuccess && sd >= 0) dev = sd; else (void)hipGetLastError();
    }
    if (dev != prev && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  }
  ~DeviceScope() { if (prev >= 0 && dev >= 0 && dev != prev) (void)hipSetDevice(prev); }
  bool ok() const { return dev >= 0; }
};
// ---- launch-time queries, answered once.  A lane-kernel launch asked the runtime five questions (device, CU count, two
// occupancy calculations, the dynamic-LDS opt-in) - a few microseconds each, every epoch, for answers that never change; the
// torch custom-op route, which spends ~30 us in the dispatcher before it gets here, was host-bound behind them.
inline int device_cu_count(int dev) {
  static std::atomic<int> memo[kMaxDev];
  if (dev < 0 || dev >= kMaxDev) return 256;
  int n = memo[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  n = 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
  memo[dev].store(n, std::memory_order_relaxed);
  return n;
}
struct LaunchMemo {
  struct Entry { const void *kern; int dev, block; size_t lds; int occ; };
  std::mutex mu;
  Entry e[64];
  int n = 0;
  const void *raised[64];        // kernels whose dynamic-LDS limit was raised to 160 KB (per device: the attribute is per code object)
  int raised_dev[64];
  int nr = 0;
};
inline LaunchMemo &launch_memo() { static LaunchMemo m; return m; }
// hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered per (kernel, device, block size, LDS bytes); 0 = the query failed
inline int occupancy_of(const void *kern, int dev, int block, size_t lds) {
  LaunchMemo &m = launch_memo();
  {
    std::lock_guard<std::mutex> lock(m.mu);
    for (int i = 0; i < m.n; ++i)
      if (m.e[i].kern == kern && m.e[i].dev == dev && m.e[i].block == block && m.e[i].lds == lds) return m.e[i].occ;
  }
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, block, lds) != hipSuccess || occ < 0) { (void)hipGetLastError(); occ = 0; }
  std::lock_guard<std::mutex> lock(m.mu);
  if (m.n < 64) m.e[m.n++] = LaunchMemo::Entry{kern, dev, block, lds, occ};
  return occ;
}
// the opt-in to more than 64 KB of dynamic LDS, once per kernel and device
inline bool raise_lds_limit(const void *kern, int dev) {
  LaunchMemo &m = launch_memo();
  {
    std::lock_guard<std::mutex> lock(m.mu);
    for (int i = 0; i < m.nr; ++i)
      if (m.raised[i] == kern && m.raised_dev[i] == dev) return true;
  }
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
  std::lock_guard<std::mutex> lock(m.mu);
  if (m.nr < 64) { m.raised[m.nr] = kern; m.raised_dev[m.nr] = dev; ++m.nr; }
  return true;
}

__device__ __forceinline__ uint32_t uni(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uni(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// k-th (0-based) set bit of a wave-uniform 64-bit word, ascending bit index:
// every lane tests its own bit, one ballot + ffs picks the winner.
__device__ __forceinline__ int kth_bit(uint64_t word, int k) {
  const int lane = lane_id();
  const bool hit = ((word >> lane) & 1ull) && ((int)__popcll(word & lanemask_lt()) == k);
  return __ffsll((unsigned long long)__ballot(hit)) - 1;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give
// each XCD one CONTIGUOUS range of work units so that the cache lines two
// neighbouring units share are fetched into one L2 only.  Pure speed: any
// placement gives the same result.
__device__ __forceinline__ int virtual_block() {
  const int nb = (int)gridDim.x, b = (int)blockIdx.x;
  const int per = nb >> 3, rem = nb & 7, x = b & 7, j = b >> 3;
  return x * per + (x < rem ? x : rem) + j;
}

// ---- single-pass prefix sums over tiles (decoupled look-back, Merrill & Garland 2016) ----------------------------
// One int64 status word per tile, written and read with 8-byte agent-scope atomics - the word IS the payload, so no
// fence pairs with it (MI355X_MICROARCH.md, inter-workgroup visibility: 8-byte agent atomics on both sides):
//   kTileEmpty (-1)   nothing published yet
//   <= -2             the tile's own sum only: -(sum) - 2
//   >= 0              the inclusive prefix up to and including the tile (final)
// Tiles are numbered by a ticket drawn when the workgroup starts, so every lower tile is running or done (HIP promises
// no dispatch order).  The words must hold kTileEmpty before the launch (a memset of 0xFF).
constexpr int64_t kTileEmpty = -1;
__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Called by ONE whole wave of the tile's workgroup with the tile's sum (sum >= 0); returns the exclusive prefix of the
// tile in every lane and leaves the inclusive prefix in the tile's word.  `word(t)` = the address of tile t's status word
// (an array of its own, or - gtok_pack_rows_scan - the slot of the output array that ends up holding that very prefix).
// The window a wave looks back through is 64 x kLookW tiles wide (kLookW words per lane, nearest first): when every tile of a
// launch is resident at once (a 250 k-row slab: ~1,000 tiles starting together) no tile has a finished prefix to offer yet,
// and a tile's wait is (its index / window) round trips - with 64-tile windows the last tiles of ZINC-full waited 15 of them.
constexpr int kLookW = 4;
template <typename W>
__device__ __forceinline__ int64_t lookback_exclusive(W word, int tile, int64_t sum) {
  const int lane = lane_id();
  if (tile == 0) {
    if (lane == 0) __hip_atomic_store(word(0), sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (lane == 0) __hip_atomic_store(word(tile), -sum - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int64_t excl = 0;
  int t = tile - 1;                                   // lane l looks at tiles t - kLookW l - k, k = 0 .. kLookW - 1
  for (;;) {
    int64_t v[kLookW];
#pragma unroll
    for (int k = 0; k < kLookW; ++k) {
      const int idx = t - lane * kLookW - k;
      v[k] = 0;                                       // below tile 0: a prefix of zero
      if (idx >= 0) v[k] = __hip_atomic_load(word(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // this lane's share: the sums of its tiles up to and including its nearest finished prefix (if it has one)
    int64_t mine = 0;
    bool has_full = false, empty_before = false;
#pragma unroll
    for (int k = 0; k < kLookW; ++k) {
      if (!has_full) {
        if (v[k] == kTileEmpty) empty_before = true;
        else if (v[k] >= 0) { mine += v[k]; has_full = true; }
        else mine += -v[k] - 2;
      }
    }
    const uint64_t full = __ballot(has_full);         // (lanes beyond tile 0 always are)
    const int p = full ? __builtin_ctzll(full) : 63;  // the nearest lane that holds a finished prefix
    const uint64_t upto = p >= 63 ? ~0ull : ((2ull << p) - 1ull);
    if (__ballot(empty_before) & upto) { __builtin_amdgcn_s_sleep(2); continue; }      // a tile in between has published nothing yet
    excl += wave_sum64(lane > p ? 0 : mine);
    if (full) break;
    t -= kWave * kLookW;
  }
  if (lane == 0) __hip_atomic_store(word(tile), excl + sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}
__device__ __forceinline__ int64_t lookback_exclusive(int64_t *__restrict__ state, int tile, int64_t sum) {
  return lookback_exclusive([state](int t) { return state + t; }, tile, sum);
}

// Philox4x32-10 (Salmon et al. 2011), the counter-based generator of the SENT spec.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: both are quarter-rate
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// trainer/train_agtt.py:171-244 (remap_zinc_tokens), branch order preserved.
__device__ __forceinline__ int remap_zinc_token(int t, int idx_off, int node_off, int edge_off) {
  if (t == 0) return 0;
  if (t >= 1 && t <= 3) return 2;
  if (t == 4) return 1;
  if (t == 5) return 2;
  if (node_off <= t && t < edge_off) {
    const int a = t - node_off;
    return (a >= 0 && a < 9) ? 8 + a : 22 + t;
  }
  if (t >= edge_off) {
    const int b = t - edge_off + 1;
    return (b >= 1 && b <= 4) ? 16 + b : 22 + t;
  }
  if (idx_off <= t && t < node_off) return 22 + (t - idx_off);
  return 22 + t;
}

// Stream one output row: ids for i < len come from f(i), the rest is pad.
// 16-byte stores (1 KiB per wave instruction) when the slab allows it.
template <typename F>
__device__ __forceinline__ void write_row(int32_t *__restrict__ row, int ld, int len, int pad, F f) {
  const int lane = lane_id();
  if (((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0)) {
    for (int i = lane * 4; i < ld; i += kWave * 4) {
      int4 v;
      v.x = (i + 0 < len) ? f(i + 0) : pad;
      v.y = (i + 1 < len) ? f(i + 1) : pad;
      v.z = (i + 2 < len) ? f(i + 2) : pad;
      v.w = (i + 3 < len) ? f(i + 3) : pad;
      *reinterpret_cast<int4 *>(row + i) = v;
    }
  } else {
    for (int i = lane; i < ld; i += kWave) row[i] = (i < len) ? f(i) : pad;
  }
}


// Same, for a row staged as uint16 tokens in LDS (8-byte aligned): 4 tokens per ds_read_b64, f(token, i) maps
// a staged token to its final id.  Reads stay below lw rounded up to 4 (the staging buffer has that slack).
template <typename F>
__device__ __forceinline__ void write_row_tok16(int32_t *__restrict__ row, int ld, int lw, int pad,
                                                const uint16_t *tok, F f) {
  const int lane = lane_id();
  if (((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0)) {
    for (int i = lane * 4; i < ld; i += kWave * 4) {
      int4 o = make_int4(pad, pad, pad, pad);
      if (i < lw) {
        const uint2 w = *reinterpret_cast<const uint2 *>(tok + i);
        o.x = f((int)(w.x & 0xFFFFu), i);
        o.y = i + 1 < lw ? f((int)(w.x >> 16), i + 1) : pad;
        o.z = i + 2 < lw ? f((int)(w.y & 0xFFFFu), i + 2) : pad;
        o.w = i + 3 < lw ? f((int)(w.y >> 16), i + 3) : pad;
      }
      *reinterpret_cast<int4 *>(row + i) = o;
    }
  } else {
    for (int i = lane; i < ld; i += kWave) row[i] = (i < lw) ? f((int)tok[i], i) : pad;
  }
}
// ... into a row of 16-bit ids (GTOK_SENT_U16): four ids per 8-byte store
template <typename F>
__device__ __forceinline__ void write_row_tok16(uint16_t *__restrict__ row, int ld, int lw, int pad,
                                                const uint16_t *tok, F f) {
  const int lane = lane_id();
  if (((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(row) & 7u) == 0)) {
    for (int i = lane * 4; i < ld; i += kWave * 4) {
      uint32_t a = (uint32_t)pad & 0xFFFFu, b = a, c = a, d = a;
      if (i < lw) {
        const uint2 w = *reinterpret_cast<const uint2 *>(tok + i);
        a = (uint32_t)f((int)(w.x & 0xFFFFu), i) & 0xFFFFu;
        if (i + 1 < lw) b = (uint32_t)f((int)(w.x >> 16), i + 1) & 0xFFFFu;
        if (i + 2 < lw) c = (uint32_t)f((int)(w.y & 0xFFFFu), i + 2) & 0xFFFFu;
        if (i + 3 < lw) d = (uint32_t)f((int)(w.y >> 16), i + 3) & 0xFFFFu;
      }
      uint2 o;
      o.x = a | (b << 16); o.y = c | (d << 16);
      *reinterpret_cast<uint2 *>(row + i) = o;
    }
  } else {
    for (int i = lane; i < ld; i += kWave) row[i] = (uint16_t)((i < lw) ? f((int)tok[i], i) : pad);
  }
}

}  // namespace gtok
