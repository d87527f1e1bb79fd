// gtok_sent.hip — SENT (Segmented Eulerian Neighbourhood Trail) walk, gfx950.
//
// Replaces autograph's Graph2TrailTokenizer.__call__ as invoked at
// trainer/train_agtt.py:250, optionally fused with remap_zinc_tokens
// (trainer/train_agtt.py:171-244) and the shortest_path query append
// (trainer/train_agtt.py:257-267).  The walk follows the spec frozen in
// DESIGN.md §SENT (upstream AutoGraph parity is unpinned); the
// bit-exact checker is oracle/gtok_oracle.c:oracle_sent.
//
// Three kernels, one spec:
//   sent_lane_kernel (gtok_sent_lane.hpp) LANE per graph (64 graphs per wave): big batches of graphs with
//                                         <= 64 nodes and <= 255 entries (every ZINC molecule)
//   sent_reg_kernel  (gtok_sent_reg.hpp)  WAVE per graph, <= 64 nodes, adjacency rows in registers (lane = node)
//   sent_lds_kernel  (gtok_sent_lds.hpp)  WAVE per graph, up to 512 nodes, immutable adjacency bit matrix in LDS
// This file is the launcher: LDS layout, kernel choice, grid sizing.
#include <cstdlib>

#include "gtok_common.hpp"
#include "gtok.h"
#include "gtok_sent_reg.hpp"
#include "gtok_sent_lds.hpp"
#include <cstdio>
#include "gtok_sent_lane.hpp"
#include "gtok_sent_blane.hpp"

namespace gtok {

// below this many graphs a launch cannot fill the chip with 64-graph waves: wave-per-graph is used instead
constexpr int GTOK_LANE_MIN_GRAPHS = 28000;     // measured crossover on ZINC-shaped molecules (lane 0.066 ms flat, reg 0.0025 ms per 1000 graphs)

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }


// below this many graphs the wave-per-graph LDS kernel is faster: the lane kernel's time is its longest walk (0.37 ms from
// 4 k to 16 k graphs of 10-256 nodes, profiles/tools/blane_crossover.sh: ER crosses at ~24 k, the family mix at ~11 k)
constexpr int GTOK_BLANE_MIN_GRAPHS = 20000;

// 0 = lane-per-graph, 1 = register-resident wave-per-graph, 2 = LDS bit matrix, 3 = lane-per-graph over the adjacency
// bit-matrix mirror.  GTOK_SENT_KERNEL=lane|reg|lds|blane pins a kernel where it is applicable (tests run every path).
static inline int sent_epochs(const gtok_sent_params *p) { return p->epoch_count > 1 ? p->epoch_count : 1; }
static inline int ncu_hint() {     // compute units of the current device (256 on MI355X)
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  return ncu > 0 ? ncu : 256;
}

static int choose_sent_kernel(const gtok_csr *g, const gtok_sent_params *p) {
  const int maxn = g->max_nodes > 0 ? g->max_nodes : 1;
  // the lane-per-graph kernels need enough WALKS to fill the chip: K epochs of a small split count like one epoch of a big one
  const int64_t walks = (int64_t)g->num_graphs * sent_epochs(p);
  const char *pin = std::getenv("GTOK_SENT_KERNEL");
  const bool pin_lane = pin && pin[0] == 'l' && pin[1] == 'a', pin_reg = pin && pin[0] == 'r';
  const bool pin_lds = pin && pin[0] == 'l' && pin[1] == 'd', pin_blane = pin && pin[0] == 'b';
  const int wneed = maxn <= 64 ? 1 : maxn <= 128 ? 2 : 4;
  const bool blane_ok = !p->labeled && !p->remap_zinc && maxn <= 256 && g->adj_rows && g->adj_planes && g->adj_words == wneed &&
                        g->adj_max_degree <= 255;
  if (pin_blane && blane_ok) return 3;
  const bool fold_ok = !p->remap_zinc || g->max_nodes <= p->max_num_nodes;   // remap folded into constants
  if (g->graph_ids || g->unit_ptr) {   // a reordered batch: the lane kernel or nothing (-1)
    const bool ok = g->graph_ids && g->unit_ptr && g->num_units > 0 && maxn <= 64 && g->max_edges <= 255 &&
                    (g->flags & GTOK_CSR_SIMPLE_SYMMETRIC) && fold_ok && g->chunk_nodes > 0;
    return ok ? 0 : -1;
  }
  const bool lane_ok = maxn <= 64 && g->max_edges <= 255 && (g->flags & GTOK_CSR_SIMPLE_SYMMETRIC) && fold_ok;
  const bool reg_ok = maxn <= 64 && g->max_edges <= 32768 && fold_ok &&
                      22 + GTOK_SENT_IDX_OFFSET + p->max_num_nodes + p->num_node_types + 256 < kEdgeRef;
  if (pin_lds) return 2;
  if (pin_lane && lane_ok) return 0;
  if (pin_reg && reg_ok) return 1;
  if (lane_ok && !pin_reg && walks >= GTOK_LANE_MIN_GRAPHS) return 0;
  if (blane_ok && !pin_reg && !pin_lane && walks >= GTOK_BLANE_MIN_GRAPHS) return 3;
  return reg_ok ? 1 : 2;
}

}  // namespace gtok

using namespace gtok;

namespace {
struct PackDest { void *packed; int64_t capacity; int64_t *row_start; int64_t *state; };   // gtok_sent_packed
}

static int sent_impl(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids, int32_t ld, int32_t *out_len,
                     const PackDest *pd, void *stream) {
  gtok::DeviceScope device_scope((hipStream_t)stream);   // the stream's device, not the thread's current one
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!g || !p || ld <= 0 || g->num_graphs < 0) return GTOK_E_INVAL;
  if (g->num_graphs == 0) return GTOK_OK;   // an empty batch is a no-op
  if (!out_ids || !out_len) return GTOK_E_INVAL;
  if (!g->node_ptr || !g->edge_ptr || !g->rowptr || (g->max_edges > 0 && !g->col)) return GTOK_E_INVAL;
  if (p->max_len < 0 || p->max_num_nodes < 0 || p->epoch_count < 0 || p->reserved != 0) return GTOK_E_INVAL;
  if (p->flags & ~(GTOK_SENT_NO_PAD | GTOK_SENT_U16 | (pd ? GTOK_SENT_PACK_ONLY : 0))) return GTOK_E_INVAL;
  const bool u16 = (p->flags & GTOK_SENT_U16) != 0;
  if (u16 && (p->pad_id < 0 || p->pad_id > 65535)) return GTOK_E_INVAL;   // (the id space itself is checked below: tokens are 16 bits inside the kernels anyway)
  const int K = sent_epochs(p);
  if ((int64_t)g->num_graphs * K > 0x7FFFFFFF / 2) return GTOK_E_TOO_LARGE;   // rows of the [K, G, ld] slab are counted in 32 bits
  if (p->labeled && (!g->nattr || !g->eattr)) return GTOK_E_INVAL;
  if (g->unit_info && !g->unit_ptr) return GTOK_E_INVAL;     // the per-unit records describe the units of unit_ptr
  if (g->max_nodes > GTOK_MAX_NODES) return GTOK_E_TOO_LARGE;
  if (g->max_edges > 60000) return GTOK_E_TOO_LARGE;   // neighbour lists are staged as uint16 in LDS
  // tokens are staged as uint16
  if ((int64_t)GTOK_SENT_IDX_OFFSET + p->max_num_nodes + p->num_node_types + 512 > 65535) return GTOK_E_TOO_LARGE;
  if (g->max_nodes + GTOK_SENT_IDX_OFFSET > 65535) return GTOK_E_TOO_LARGE;

  const int maxn = g->max_nodes > 0 ? g->max_nodes : 1;
  const int W = maxn <= 64 ? 1 : maxn <= 128 ? 2 : maxn <= 256 ? 4 : 8;
  const int cap = p->max_len < ld ? p->max_len : ld;
  const int maxe = g->max_edges > 0 ? g->max_edges : 1;
  const int which = choose_sent_kernel(g, p);
  if (which < 0) return GTOK_E_INVAL;       // a reordered batch the lane-per-graph kernel cannot take
  if (pd) {   // rows appended to a packed buffer by the walk itself: the lane-per-graph molecule kernel, rows that start on 16-byte boundaries
    if (which != 0) return GTOK_E_UNSUPPORTED;
    if (!pd->packed || !pd->row_start || !pd->state || pd->capacity < 0) return GTOK_E_INVAL;
    if (ld > (u16 ? 2040 : 1020)) return GTOK_E_TOO_LARGE;    // (a row's 16-byte pieces are counted in 8 bits)
    if (ld % (u16 ? 8 : 4) != 0 || (reinterpret_cast<uintptr_t>(out_ids) & 15u) || (reinterpret_cast<uintptr_t>(pd->packed) & 15u)) return GTOK_E_INVAL;
  }
  // padding with non-temporal stores once the slab outgrows the memory-side cache (256 MB on MI355X): gtok_sent_lane.hpp
  int pad_nt = (int64_t)g->num_graphs * K * ld * (u16 ? 2 : 4) > ((int64_t)256 << 20);
  if (const char *cs = std::getenv("GTOK_PAD_NT")) pad_nt = cs[0] != '0';   // tuning knob
  const bool lane_path = which == 0, reg_path = which == 1;
  if (which == 3) {
    SentBLaneArgs a;
    a.g = *g; a.p = *p; a.out = out_ids; a.ld = ld; a.out_len = out_len;
    a.units = (g->num_graphs + 63) / 64;
    a.epochs = K;
    a.pad_nt = pad_nt;
    a.prio = 1;
    if (const char *cs = std::getenv("GTOK_BLANE_PRIO")) a.prio = cs[0] != '0';   // tuning knob
    const bool p4 = g->adj_max_degree <= 15;
    typedef void (*KF)(const SentBLaneArgs);
#define GTOK_BLANE_K(U)                                                                                 \
  (W == 1 ? (p4 ? (KF)sent_blane_kernel<1, 4, U> : (KF)sent_blane_kernel<1, 8, U>)                     \
   : W == 2 ? (p4 ? (KF)sent_blane_kernel<2, 4, U> : (KF)sent_blane_kernel<2, 8, U>)                   \
            : (p4 ? (KF)sent_blane_kernel<4, 4, U> : (KF)sent_blane_kernel<4, 8, U>))
    KF kern = u16 ? GTOK_BLANE_K(true) : GTOK_BLANE_K(false);
#undef GTOK_BLANE_K
    const size_t lds = (size_t)5120 * W;   // 20 W dwords per lane (gtok_sent_blane.hpp)
    const int dev = device_scope.dev, ncu = device_cu_count(dev);
    // one workgroup per CU: 8 waves at W = 4 (160 KB of LDS, 2 per SIMD: the register file holds no more), else 16
    int nw = W == 4 ? 8 : 16;
    if (const char *cs = std::getenv("GTOK_BLANE_WAVES")) { const int c = std::atoi(cs); if (c == 4 || c == 8 || c == 16) nw = c < nw ? c : nw; }   // tuning knob
    if (!raise_lds_limit(reinterpret_cast<const void *>(kern), dev)) return GTOK_E_LAUNCH;
    const int vun = a.units * K;
    const int nb = vun < ncu ? vun : ncu;   // every CU, also when some of its waves stay without a unit
    if (2 * nb > kSlotInts) return GTOK_E_TOO_LARGE;      // (two ticket words per workgroup in one counter block: 528 CUs)
    // unit-major always: the K walks of a unit fetch the same adjacency rows at about the same time (L2 hits on the one HBM read
    // of a step) - 125 k ER graphs x 8 epochs: 0.356 ms per epoch against 0.404 epoch-major (0.49 for one epoch per launch)
    a.epoch_major = 0;
    if (const char *cs = std::getenv("GTOK_LANE_PAIR_ORDER")) a.epoch_major = cs[0] == 'e';   // tuning knob: unit | epoch
    // the ticket counters of the pairs beyond the first round: a block of this launch's own, in device memory, zero between launches
    QueueSlot slot = take_queue_slot(dev, (hipStream_t)stream);
    if (!slot.counters) return slot.graph_pool_empty ? GTOK_E_GRAPH_SLOTS : GTOK_E_LAUNCH;
    a.tickets = slot.counters;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * nw), lds * nw, (hipStream_t)stream, a);
    mark_queue_slot(slot, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
  }
  if (lane_path) {
    SentLaneArgs a;
    a.g = *g; a.p = *p; a.maxn = maxn; a.out = out_ids; a.ld = ld; a.out_len = out_len;
    a.pack_out = pd ? pd->packed : nullptr;
    a.pack_start = pd ? pd->row_start : nullptr;
    a.pack_state = pd ? reinterpret_cast<unsigned long long *>(pd->state) : nullptr;
    a.pack_region_cap = 0; a.pack_regions = 1;
    a.pack_scratch = pd && (p->flags & GTOK_SENT_PACK_ONLY) ? 1 : 0;
    if (a.pack_scratch) a.p.flags |= GTOK_SENT_NO_PAD;                   // (nobody reads the staging rows' tails)
    // staging sized by the largest 64-graph chunk when the host told us, else by the per-graph maxima
    a.cap_n = g->chunk_nodes > 0 ? g->chunk_nodes : 64 * maxn;
    a.cap_e = g->chunk_edges > 0 ? g->chunk_edges : 64 * maxe;
    a.cap_r = a.cap_n + 64;
    int off = 0;
    a.off_rp = off; off += align_up(a.cap_r + 16, 16);  // + 16: the counter set-up reads whole dwords around a lane's row pointers
    a.off_col = off; off += align_up(a.cap_e + 8, 16);   // +8: a row's first four entries are read as an aligned dword pair
    a.off_eat = a.off_nat = off;
    if (p->labeled) {
      a.off_eat = off; off += align_up(a.cap_e + 8, 16);
      a.off_nat = off; off += align_up(a.cap_n + 8, 16);
    } else {
      a.off_nat = off; off += align_up(a.cap_n + 8, 16);  // unlabelled: the node -> visit index table alone
    }
    a.lds = align_up(off, 16);
    if (a.lds < 512) a.lds = 512;       // (a wave's slice also carries its last unit's 64 (row, pad start) pairs: the shared padding)
    if (a.lds > 64 * 1024) return GTOK_E_TOO_LARGE;
    // counter planes: degree < 2^P; an unknown max_degree is covered by P = 6 (a node of a simple graph with <= 64 nodes has < 64 neighbours)
    const bool p3 = g->max_degree > 0 && g->max_degree <= 15;
    typedef void (*KF)(const SentLaneArgs);
    const bool pk = g->rowptr8 && g->col8;
#define GTOK_LANE_K2(LAB, REMAP, U)                                                                                   \
  (pk ? (p3 ? (KF)sent_lane_kernel<LAB, 4, REMAP, true, U> : (KF)sent_lane_kernel<LAB, 6, REMAP, true, U>)            \
      : (p3 ? (KF)sent_lane_kernel<LAB, 4, REMAP, false, U> : (KF)sent_lane_kernel<LAB, 6, REMAP, false, U>))
#define GTOK_LANE_K(LAB, REMAP) (u16 ? GTOK_LANE_K2(LAB, REMAP, true) : GTOK_LANE_K2(LAB, REMAP, false))
    KF kern = !p->labeled ? GTOK_LANE_K(false, false) : p->remap_zinc ? GTOK_LANE_K(true, true) : GTOK_LANE_K(true, false);
#undef GTOK_LANE_K
#undef GTOK_LANE_K2
    if (pd) {   // gtok_sent_packed: the instantiations that append to the packed buffer (batches with the byte mirror only)
      if (!pk) return GTOK_E_UNSUPPORTED;
#define GTOK_LANE_KP2(LAB, REMAP, U) (p3 ? (KF)sent_lane_kernel<LAB, 4, REMAP, true, U, true> : (KF)sent_lane_kernel<LAB, 6, REMAP, true, U, true>)
#define GTOK_LANE_KP(LAB, REMAP) (u16 ? GTOK_LANE_KP2(LAB, REMAP, true) : GTOK_LANE_KP2(LAB, REMAP, false))
      kern = !p->labeled ? GTOK_LANE_KP(false, false) : p->remap_zinc ? GTOK_LANE_KP(true, true) : GTOK_LANE_KP(true, false);
#undef GTOK_LANE_KP
#undef GTOK_LANE_KP2
    }
    const int dev = device_scope.dev, ncu = device_cu_count(dev);
    int occ = -1;                                 // occupancy of one-wave workgroups: only the fallback launch below asks
    if (const char *cs = std::getenv("GTOK_LANE_BLOCKS_PER_CU")) {   // tuning knob
      occ = occupancy_of(reinterpret_cast<const void *>(kern), dev, 64, (size_t)a.lds);
      if (occ < 1) occ = 1;
      const int c = std::atoi(cs);
      if (c >= 1 && c < occ) occ = c;
    }
    a.units = g->unit_ptr ? g->num_units : (g->num_graphs + 63) / 64;
    a.epochs = K;
    a.pad_nt = pad_nt;
    a.epoch_major = 0;     // set below, once the launch shape is known
    if ((int64_t)a.units * K > 0x7FFFFFFF / 64) return GTOK_E_TOO_LARGE;
    const int vunits = a.units * K;               // (unit, epoch) pairs the launch walks
    if (pd) {   // regions of the packed buffer: as many as leave every region >= 64 pairs (they fill evenly: pairs are dealt by rank in walk length)
      int m = GTOK_PACK_REGIONS;
      while (m > 1 && vunits < 64 * m) m >>= 1;
      a.pack_regions = m;
      a.pack_region_cap = (pd->capacity / m) & ~(int64_t)7;
    }
    // order of the pairs: unit-major (the K walks of a unit side by side: the deal stays sorted by walk length) for splits that
    // need several epochs to fill the chip; epoch-major (epoch 0 of every unit, then epoch 1, ...) when one epoch fills half of
    // the resident waves or more - the first round is then the tuned one-epoch deal (249,456 molecules x 2 epochs, int32 padded:
    // 0.076 ms per epoch against 0.089; x 4 as 16-bit rows: 0.0573 against 0.0589; 12 k x 24: 0.0043 against 0.0039 the other way)
    a.epoch_major = K > 1 && a.units >= 8 * ncu;
    if (const char *cs = std::getenv("GTOK_LANE_PAIR_ORDER")) a.epoch_major = cs[0] == 'e';   // tuning knob: unit | epoch
    a.unit_mul = 0;
    a.prio_cut[0] = 16; a.prio_cut[1] = 32; a.prio_cut[2] = 48;   // quartiles (profiles/tools/lane_prio_sweep.sh)
    if (const char *pc = std::getenv("GTOK_LANE_PRIO_CUTS")) std::sscanf(pc, "%d,%d,%d", &a.prio_cut[0], &a.prio_cut[1], &a.prio_cut[2]);   // tuning knob
    // a reordered batch: one 16-wave workgroup per CU (the kernel pairs long units with short ones on every SIMD) when a
    // wave's share of the CU's 160 KB is enough and the batch fills the chip; else one-wave workgroups with the units spread
    const char *wg = std::getenv("GTOK_LANE_PER_CU");   // tuning knob: 0 = never
    // (+ kLaneWgShared bytes: the workgroup's ticket counter, from which its waves draw their units after the first round,
    // and the words through which they share the padding of their last units)
    if (g->unit_ptr && a.lds * 16 + kLaneWgShared <= 160 * 1024 && vunits >= 4 * ncu && !(wg && wg[0] == '0')) {
      // (the opt-in to more than 64 KB of dynamic LDS is per kernel and per device: a host-side call of a microsecond)
      const bool r = raise_lds_limit(reinterpret_cast<const void *>(kern), dev);
      int occ16 = 0;
      // One 16-wave workgroup per CU, except where ONE nearly full round of units ends in the int32 slab's padding (ZINC-full x 1
      // epoch, a 31 k shard x 8): there two 8-wave workgroups per CU let one half of a CU pad while the other still walks
      // (0.0740 against 0.0769 ms, 8.98 against 9.50 us per epoch; every other shape - fewer units, several rounds, 16-bit rows,
      // no padding - is 1-4 % better with 16: profiles/r04/wg_waves.txt)
      const int slots = ncu * 16;
      const bool one_full_round = vunits <= slots && 4 * (int64_t)vunits > 3 * (int64_t)slots;
      const char *ww = std::getenv("GTOK_LANE_WG_WAVES");   // tuning knob: 8 = two 8-wave workgroups per CU
      const int wgw = ww ? ((ww[0] == '8') ? 8 : (ww[0] == '1' && ww[1] == '2') ? 12 : 16)   // 12: three waves per SIMD
                         : (one_full_round && !u16 && !(p->flags & GTOK_SENT_NO_PAD) ? 8 : 16);
      const int per_cu = wgw == 8 ? 2 : 1;
      const size_t wg_lds = (size_t)a.lds * wgw + kLaneWgShared;
      if (r && (occ16 = occupancy_of(reinterpret_cast<const void *>(kern), dev, 64 * wgw, wg_lds)) >= per_cu) {
        hipLaunchKernelGGL(kern, dim3(ncu * per_cu), dim3(64 * wgw), wg_lds, (hipStream_t)stream, a);
        return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
      }
      (void)hipGetLastError();
    }
    if (g->unit_ptr && vunits > 16) {
      auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
      int m = (int)(vunits * 0.6180339887498949);
      while (gcd(m, vunits) != 1) ++m;
      a.unit_mul = m;
    }
    if (occ < 0) {
      occ = occupancy_of(reinterpret_cast<const void *>(kern), dev, 64, (size_t)a.lds);
      if (occ < 1) occ = 1;
    }
    int nb = ncu * occ;
    if (nb > vunits) nb = vunits;
    if (a.pack_scratch && nb > ncu * 16) nb = ncu * 16;     // (the staging space is 64 rows for each of 16 waves per CU)
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64), (size_t)a.lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
  }
  SentArgs a;
  a.g = *g; a.p = *p; a.cap = cap; a.maxn = maxn; a.out = out_ids; a.ld = ld; a.out_len = out_len;
  a.epochs = K;
  const int pairs = g->num_graphs * K;          // (epoch, graph) pairs the launch walks
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
    const int64_t bound = p->labeled ? 2 + 7 * (int64_t)maxn + 2 * (int64_t)maxe : 2 + 5 * (int64_t)maxn + (int64_t)maxe;
    const int tokcap = (int)(bound < p->max_len ? bound : p->max_len) + kSentSlack + 2 * maxn;
    a.l.adj = off; off += maxn * W * 8;
    a.l.vis = a.l.rng = a.l.vidx = 0;
    a.l.order = off; off += align_up(maxn * 2, 8);
    a.l.tok = off; off += align_up(tokcap * 2, 8);
    a.l.col = a.l.rp = a.l.eat = a.l.nat = off;
    if (p->labeled) {
      a.l.col = off; off += align_up(maxe * 2, 8);
      a.l.rp = off; off += align_up((maxn + 1) * 4, 8);
      a.l.eat = off; off += align_up(maxe, 8);
      a.l.nat = off; off += align_up(maxn, 8);
    }
  }
  a.l.stride = align_up(off, 16);
  if (a.l.stride > 160 * 1024) return GTOK_E_TOO_LARGE;
  // sent_lds_kernel waves never cooperate and draw graphs from a queue: one wave per workgroup packs LDS best
  int wpb = reg_path ? 4 : 1;
  while (wpb > 1 && wpb * a.l.stride > 64 * 1024) wpb >>= 1;
  const size_t lds = (size_t)wpb * a.l.stride;

  void (*kern)(const SentArgs) = nullptr;
#define PICK(w)                                                                   \
  kern = p->labeled ? (void (*)(const SentArgs))sent_lds_kernel<w, true>         \
                    : (void (*)(const SentArgs))sent_lds_kernel<w, false>
  if (reg_path) {
    // no truncation test in the walk when max_len can hold the longest possible trail of this batch
    const int64_t bound = p->labeled ? 2 + 7 * (int64_t)maxn + 2 * (int64_t)maxe : 2 + 5 * (int64_t)maxn + (int64_t)maxe;
    const bool nolim = bound <= p->max_len;
    typedef void (*KF)(const SentArgs);
    kern = p->labeled ? (nolim ? (KF)sent_reg_kernel<true, true> : (KF)sent_reg_kernel<true, false>)
                      : (nolim ? (KF)sent_reg_kernel<false, true> : (KF)sent_reg_kernel<false, false>);
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
  a.queue = nullptr;
  QueueSlot slot;
  int nb;
  if (reg_path) {
    occ = gtok::resident_blocks(occ);
    a.units = (pairs + wpb - 1) / wpb;
    nb = ncu * occ;
    if (nb > a.units) nb = a.units;
    a.upb = (a.units + nb - 1) / nb;
    nb = (a.units + a.upb - 1) / a.upb;
  } else {
    occ = gtok::resident_waves(occ);     // one-wave workgroups
    a.units = pairs; a.upb = 0;
    nb = ncu * occ;
    if (nb > a.units) nb = a.units;
    slot = take_queue_slot(dev, (hipStream_t)stream);
    a.queue = slot.counters;
    if (!a.queue) return slot.graph_pool_empty ? GTOK_E_GRAPH_SLOTS : GTOK_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(nb), dim3(wpb * 64), lds, (hipStream_t)stream, a);
  mark_queue_slot(slot, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// adjacency bit-matrix mirror (include/gtok.h: gtok_csr_adjbits)
// ---------------------------------------------------------------------------------------------
extern "C" int gtok_sent(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids,
                         int32_t ld, int32_t *out_len, void *stream) {
  return sent_impl(g, p, out_ids, ld, out_len, nullptr, stream);
}

extern "C" int gtok_sent_packed(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids, int32_t ld, int32_t *out_len,
                                void *packed, int64_t capacity, int64_t *row_start, int64_t *state, void *stream) {
  if (!g || !p) return GTOK_E_INVAL;
  if (g->num_graphs == 0) return GTOK_OK;
  const PackDest pk{packed, capacity, row_start, state};
  return sent_impl(g, p, out_ids, ld, out_len, &pk, stream);
}

extern "C" int64_t gtok_sent_pack_scratch_rows(void *stream) {
  gtok::DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  return (int64_t)64 * 16 * device_cu_count(device_scope.dev);
}

extern "C" int gtok_csr_adjbits(const gtok_csr *g, int32_t words, uint64_t *rows, uint64_t *planes, int32_t *info, void *stream) {
  gtok::DeviceScope device_scope((hipStream_t)stream);
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!g || g->num_graphs < 0 || (words != 1 && words != 2 && words != 4) || g->graph_ids || g->unit_ptr) return GTOK_E_INVAL;
  if (g->max_nodes > 64 * words) return GTOK_E_TOO_LARGE;
  if (g->num_graphs == 0) return GTOK_OK;
  if (!g->node_ptr || !g->edge_ptr || !g->rowptr || (g->max_edges > 0 && !g->col) || !rows || !planes || !info) return GTOK_E_INVAL;
  AdjBitsArgs a;
  a.g = *g; a.W = words; a.rows = rows; a.planes = planes; a.info = info;
  const int wpb = words == 4 ? 2 : 4;                         // 64 W^2 words of LDS per wave: 8 KB at W = 4
  const size_t lds = (size_t)wpb * 64 * words * words * 8;
  int nb = (g->num_graphs + wpb - 1) / wpb;
  if (nb > 256 * 8) nb = 256 * 8;
  typedef void (*K)(const AdjBitsArgs);
  K kern = words == 1 ? (K)adj_bits_kernel<1> : words == 2 ? (K)adj_bits_kernel<2> : (K)adj_bits_kernel<4>;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(wpb * 64), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// byte-packed mirror of rowptr / col (include/gtok.h: gtok_csr_pack8)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) csr_pack8_kernel(const int32_t *__restrict__ src, int64_t n, uint8_t *__restrict__ dst) {
  const int64_t nv = n >> 2, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += step) {
    const I32x4 v = reinterpret_cast<const I32x4 *>(src)[t];
    const uint32_t w = ((uint32_t)v.x & 255u) | (((uint32_t)v.y & 255u) << 8) | (((uint32_t)v.z & 255u) << 16) | ((uint32_t)v.w << 24);
    __builtin_memcpy(dst + 4 * t, &w, 4);
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) dst[(nv << 2) + threadIdx.x] = (uint8_t)src[(nv << 2) + threadIdx.x];
}

extern "C" int gtok_csr_pack8(const gtok_csr *g, int64_t num_rowptr, int64_t num_col, uint8_t *rowptr8, uint8_t *col8, void *stream) {
  gtok::DeviceScope device_scope((hipStream_t)stream);   // the stream's device, not the thread's current one
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (!g || num_rowptr < 0 || num_col < 0) return GTOK_E_INVAL;
  if (g->max_edges > 255 || g->max_nodes > 256) return GTOK_E_TOO_LARGE;
  if ((num_rowptr && (!g->rowptr || !rowptr8)) || (num_col && (!g->col || !col8))) return GTOK_E_INVAL;
  auto run = [&](const int32_t *src, int64_t n, uint8_t *dst) {
    if (n == 0) return;
    const int64_t nb = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(csr_pack8_kernel, dim3((unsigned)(nb < 1 ? 1 : nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, src, n, dst);
  };
  run(g->rowptr, num_rowptr, rowptr8);
  run(g->col, num_col, col8);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// SENT decoder: token rows -> graphs in visit-index space (DESIGN.md section 5 read backwards)
// ---------------------------------------------------------------------------------------------
// Lane per row (the stream is a sequential grammar): position tokens in first-visit order name nodes 0, 1, 2, ...;
// a position token after another one is a trail edge; `LADJ [type] pos ... RADJ` lists edges from the current node
// to earlier ones; RESET breaks the trail; labelled rows carry an edge-type token ahead of every trail step and a
// node-type token after every first visit.  Output is the edge list (a = the node the edge was written from, b = the
// other end, t = edge type or -1), the node types, and a status: 0 complete (EOS reached), 1 malformed, 2 an
// output capacity exceeded, 3 well-formed but cut before EOS (a row truncated at max_len).
struct DecodeArgs {
  const int32_t *ids; int ld; const int32_t *len; int rows;
  int idx_off, node_off, edge_off, labeled;
  int32_t *num_nodes, *num_edges, *edge_a, *edge_b, *edge_t, *node_type, *status;
  int ecap, ncap;
};

__global__ void __launch_bounds__(256) sent_decode_kernel(const DecodeArgs a) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (g >= a.rows) return;
  const int32_t *__restrict__ t = a.ids + (int64_t)g * a.ld;
  const int L = min(a.len[g], a.ld);
  int32_t *ea = a.edge_a + (int64_t)g * a.ecap, *eb = a.edge_b + (int64_t)g * a.ecap, *et = a.edge_t + (int64_t)g * a.ecap;
  int32_t *nt = a.node_type + (int64_t)g * a.ncap;
  // a full output array does not stop the parse (include/gtok.h: with status 2 "counts are still right"): what does not
  // fit is dropped, the row is read to its end, and only a row that is otherwise fine (0 or 3) reports 2 - so edge_cap =
  // node_cap = 0 is a count-only pass (bench.py: nodes a truncated walk reached)
  int st = 3, prev = -1, nseen = 0, pending = -1, m = 0, i = 1;
  bool over = false;
  auto add = [&](int from, int to, int type) {
    if (m < a.ecap) { ea[m] = from; eb[m] = to; et[m] = type; } else over = true;
    ++m;
  };
  if (L < 1 || t[0] != GTOK_SENT_SOS) st = 1;
  while (st == 3 && i < L) {
    const int tk = t[i];
    if (tk == GTOK_SENT_EOS) { st = 0; break; }
    if (tk == GTOK_SENT_RESET) { prev = -1; pending = -1; ++i; continue; }
    if (tk == GTOK_SENT_LADJ) {
      if (prev < 0) { st = 1; break; }
      ++i;
      while (i < L && t[i] != GTOK_SENT_RADJ) {
        int type = -1;
        if (a.labeled) { type = t[i] - a.edge_off; if (type < 0) { st = 1; break; } if (++i >= L) break; }
        const int b = t[i] - a.idx_off;
        if (b < 0 || b >= nseen) { st = 1; break; }
        add(prev, b, type);
        ++i;
      }
      if (st == 1 || i >= L) break;           // malformed, or cut inside the bracket
      ++i;
      continue;
    }
    const bool is_pos = tk >= a.idx_off && tk < a.node_off;
    if (a.labeled && !is_pos) {               // edge type ahead of a trail step
      if (tk < a.edge_off) { st = 1; break; }
      pending = tk - a.edge_off; ++i; continue;
    }
    if (!is_pos) { st = 1; break; }
    const int k = tk - a.idx_off;
    ++i;
    if (k == nseen) {                         // first visit
      int type = -1;
      if (a.labeled) {
        if (i >= L) { if (nseen < a.ncap) nt[nseen] = -1; else over = true; ++nseen; break; }   // cut between the position and its type
        type = t[i] - a.node_off;
        if (type < 0) { st = 1; break; }      // (a type beyond num_node_types aliases the edge range: the grammar decides)
        ++i;
      }
      if (nseen < a.ncap) nt[nseen] = type; else over = true;
      ++nseen;
    } else if (k > nseen) { st = 1; break; }
    if (prev >= 0) add(prev, k, pending);
    pending = -1;
    prev = k;
  }
  if (over && st != 1) st = 2;
  a.num_nodes[g] = nseen; a.num_edges[g] = m; a.status[g] = st;
}

extern "C" int gtok_sent_decode(const int32_t *ids, int32_t ld, const int32_t *len, int32_t num_rows,
                                int32_t max_num_nodes, int32_t labeled, int32_t num_node_types,
                                int32_t *num_nodes, int32_t *num_edges, int32_t *edge_a, int32_t *edge_b, int32_t *edge_type,
                                int32_t edge_cap, int32_t *node_type, int32_t node_cap, int32_t *status, void *stream) {
  gtok::DeviceScope device_scope((hipStream_t)stream);   // the stream's device, not the thread's current one
  if (!device_scope.ok()) return GTOK_E_NO_DEVICE;
  if (num_rows < 0 || ld <= 0 || edge_cap < 0 || node_cap < 0 || max_num_nodes < 0) return GTOK_E_INVAL;
  if (num_rows == 0) return GTOK_OK;
  if (!ids || !len || !num_nodes || !num_edges || !status) return GTOK_E_INVAL;
  if ((edge_cap > 0 && (!edge_a || !edge_b || !edge_type)) || (node_cap > 0 && !node_type)) return GTOK_E_INVAL;   // (a capacity of 0: count only, no array)
  DecodeArgs a;
  a.ids = ids; a.ld = ld; a.len = len; a.rows = num_rows; a.labeled = labeled;
  a.idx_off = GTOK_SENT_IDX_OFFSET; a.node_off = a.idx_off + max_num_nodes; a.edge_off = a.node_off + num_node_types;
  a.num_nodes = num_nodes; a.num_edges = num_edges; a.edge_a = edge_a; a.edge_b = edge_b; a.edge_t = edge_type;
  a.node_type = node_type; a.status = status; a.ecap = edge_cap; a.ncap = node_cap;
  hipLaunchKernelGGL(sent_decode_kernel, dim3((num_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? GTOK_OK : GTOK_E_LAUNCH;
}

extern "C" const char *gtok_sent_kernel_name(const gtok_csr *g, const gtok_sent_params *p) {
  if (!g || !p) return "";
  static const char *lds[] = {"sent_lds_kernel<W=1>", "sent_lds_kernel<W=2>", "sent_lds_kernel<W=4>", "sent_lds_kernel<W=8>"};
  const int maxn = g->max_nodes > 0 ? g->max_nodes : 1;
  static const char *bl[] = {"sent_blane_kernel<W=1>", "sent_blane_kernel<W=2>", "sent_blane_kernel<W=4>"};
  switch (choose_sent_kernel(g, p)) {
    case -1: return "";
    case 0: return "sent_lane_kernel";
    case 1: return "sent_reg_kernel";
    case 3: return bl[maxn <= 64 ? 0 : maxn <= 128 ? 1 : 2];
    default: return lds[maxn <= 64 ? 0 : maxn <= 128 ? 1 : maxn <= 256 ? 2 : 3];
  }
}
