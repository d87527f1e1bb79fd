/*
 * gtok.h — C ABI of libgtok.so, the MI355X (gfx950) graph->sequence tokenizer.
 *
 * Every entry point replaces one piece of the reference's Python hot path
 * (paths relative to the reference checkout):
 *
 *   gtok_ibtt_zinc     graph_data_loader/zinc_dataset_indexbase.py:143-227
 *                      (tokenize_molecule + __getitem__ truncation) fused with
 *                      graph_data_loader/data_loader.py:465-484 (TokenDataset:
 *                      strip after <p>, vocab lookup, [:max_len])
 *   gtok_text_to_ids   graph_data_loader/data_loader.py:465-484 (TokenDataset on
 *                      arbitrary whitespace-tokenised text, any grammar)
 *   gtok_ibtt_synth    the graph-token index-based grammar
 *                      (docs/synthetic_data.md:46-68) emitted straight from the
 *                      edge list, then data_loader.py:479-482
 *   gtok_sent          autograph Graph2TrailTokenizer.__call__ as called at
 *                      trainer/train_agtt.py:250 (SENT walk; spec in DESIGN.md,
 *                      upstream parity unpinned), optionally fused with
 *                      remap_zinc_tokens (:171-244) and the query append (:257-267)
 *   gtok_sent_packed   gtok_sent whose walk also appends every row to a packed buffer (the payload of the all-gather behind the
 *                      val / test loaders, trainer/train_agtt.py:602-607, and of the D2H copy behind :246-273) - no second pass
 *   gtok_sent_decode   the SENT spec read backwards (token rows -> graphs in visit-index space)
 *   gtok_remap_zinc    trainer/train_agtt.py:171-244 on an existing token slab
 *   gtok_collate       data_loader.py:488-497 and trainer/train_agtt.py:276-302
 *                      (gather rows of a batch, pad to the batch max, bool mask)
 *   gtok_parse_graph_text  graph_token_dataset_autograph.py:14-158 (text -> edges, query, label)
 *   gtok_find_token    the `<q>` search of trainer/train_ibtt.py:88-103 on a collated batch
 *   gtok_row_offsets / gtok_pack_rows / gtok_pack_rows_u16 / gtok_pack_rows_scan / gtok_unpack_rows / gtok_unpack_rows_checked / gtok_unpack_rows_u16 / gtok_unpack_rows_at / gtok_collate_packed
 *                      (no reference counterpart) the packed form of a token slab - rows back to back, 16 or 32
 *                      bits per id - for the copies that leave the GPU: the all-gather that reassembles the rows of
 *                      every rank in dataset order (val/test loaders, trainer/train_agtt.py:602-607) and the D2H
 *                      copy behind TokenizedGraphDataset.__getitem__ (trainer/train_agtt.py:246-273); the readers also
 *                      take a [rows, ld] slab of 16-bit ids in place (row_ptr NULL): the per-batch collate of
 *                      trainer/train_agtt.py:276-302 straight over GTOK_SENT_U16 rows
 *   gtok_collate_epoch_plan / gtok_collate_epoch   trainer/train_agtt.py:276-302 for every batch of an epoch at once
 *   gtok_collate_batch  the same for one batch whose row list is a host array (what torch's DataLoader hands over)
 *   gtok_ids_to_text   graph_data_loader/zinc_dataset_indexbase.py:143-227, the STRING form (ids rendered through a string table)
 *   gtok_zinc_text_tails   zinc_dataset_indexbase.py:186-195, :217-221 (label token + <eos> / the max_len cut, per molecule)
 *   gtok_csr_check     (no reference counterpart; the property torch_geometric's coalesced undirected graphs have by construction -
 *                      zinc_dataset_autograph.py:51-73 passes them through) verifies GTOK_CSR_SIMPLE_SYMMETRIC on the device and
 *                      measures max_nodes / max_edges / max_degree
 *   gtok_csr_lane_sort (no reference counterpart) the reordered copy of a small-graph batch that the lane-per-graph SENT kernel
 *                      walks fastest (graph_ids / unit_ptr / unit_info + the permuted arrays and their byte mirror), built on
 *                      the device: what a caller of trainer/train_agtt.py:246-250's tokenizer prepares once per split
 *   gtok_csr_pack8     (no reference counterpart) byte-packed mirror of the CSR index arrays of small-graph batches
 *   gtok_csr_adjbits   (no reference counterpart) adjacency bit-matrix mirror of batches of graphs with <= 256 nodes
 *   gtok_vocab_stats_text   the corpus pass of build_vocab_from_texts / the ZINC dynamic-token scan over arbitrary texts
 *   gtok_vocab_stats_synth  the corpus pass of build_vocab_from_texts
 *                      (data_loader.py:451-463) for graph-token corpora held as
 *                      CSR: per node-id token, occurrence count and first position
 *
 * Conventions: all pointers are DEVICE pointers borrowed from the caller
 * (never freed or retained), `stream` is a hipStream_t passed as void*, the
 * call only enqueues work on that stream (no synchronisation; the only allocation
 * is a small per-device block of work-queue counters on a device's FIRST launch,
 * so warm up once before capturing a hipGraph), outputs are caller-allocated.  Return value is
 * 0 or a negative GTOK_E_* code; nothing throws across the ABI.  Re-entrant.
 *
 * Output convention of every tokenizer entry point: out_ids is a row-major
 * [rows, ld] int32 slab; row g receives its first min(out_len[g], ld) ids and
 * pad_id after them; out_len[g] is the TRUE sequence length (already cut at
 * max_len as the reference does), so out_len[g] > ld tells the caller that
 * the slab was too narrow for that row.  gtok_sent only (ABI v4): epoch_count = K
 * makes that K slabs back to back ([K, G, ld], out_len [K, G]); GTOK_SENT_U16
 * makes the ids 16 bits wide; GTOK_SENT_NO_PAD leaves the pad tails unwritten.
 *
 * Batched CSR layout (gtok_csr), G graphs, graph g has N_g nodes and E_g
 * directed adjacency entries exactly as the source edge_index lists them:
 *   node_ptr[G+1]  int32  prefix sum of N_g
 *   edge_ptr[G+1]  int64  prefix sum of E_g
 *   rowptr[sum(N_g)+G] int32  graph g's N_g+1 LOCAL row pointers start at
 *                             node_ptr[g]+g; entry k of row u lives at
 *                             edge_ptr[g] + rowptr_g[u] + k
 *   col[sum E_g]   int32  local neighbour id
 *   eorder[sum E_g] int32 position of the entry in the graph's original COO
 *                         edge list (NULL = identity, i.e. edge_index was
 *                         already row-sorted); IBTT's first-occurrence rule
 *                         needs it
 *   nattr[sum N_g] uint8  node type (x); NULL for unlabelled graphs
 *   eattr[sum E_g] uint8  edge type (edge_attr); NULL for unlabelled graphs
 */
#ifndef GTOK_H
#define GTOK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTOK_OK 0
#define GTOK_E_INVAL (-1)     /* null pointer / negative size / bad flag        */
#define GTOK_E_TOO_LARGE (-2) /* a graph exceeds the LDS-resident limits        */
#define GTOK_E_LAUNCH (-3)    /* hipLaunchKernel reported an error              */
#define GTOK_E_NO_DEVICE (-4) /* no gfx950 device visible                       */
#define GTOK_E_GRAPH_SLOTS (-5) /* a launch captured into a hipGraph needed a reserved block of work-queue counters and all 64
                                   per device are held by LIVE graphs: the ticket-scheduled kernels ("sent_lds_kernel",
                                   "sent_blane_kernel", "ibtt_zinc_lane_kernel", gtok_pack_rows_scan) keep one per captured launch for as long as the capturing graph
                                   (and the executable graphs made from it) exist - destroying a graph returns its blocks   */

#define GTOK_E_UNSUPPORTED (-6) /* the request is valid but not for the kernel this batch runs (gtok_sent_packed on a batch that
                                   gtok_sent_kernel_name() does not answer "sent_lane_kernel" for): use the two-call route    */

#define GTOK_MAX_NODES 512 /* SENT adjacency bit-matrix rows per wave (LDS)  */

/* flags: properties the CALLER has verified for the whole batch (on the host, or with gtok_csr_check on the device) */
#define GTOK_CSR_SIMPLE_SYMMETRIC 1 /* simple undirected graphs, both directions stored: no self-loop, no entry
                                       listed twice, and (v,u) is listed whenever (u,v) is - every row is the
                                       node's complete neighbour list (PyG-coalesced molecules).  A batch flagged
                                       wrongly gets wrong tokens (never an out-of-bounds access).                */

typedef struct gtok_csr {
  int32_t num_graphs;
  int32_t max_nodes; /* max N_g over the batch (host-known)                 */
  int32_t max_edges; /* max E_g over the batch (host-known)                 */
  int32_t flags;     /* GTOK_CSR_* (0 = nothing known)                      */
  const int32_t *node_ptr;
  const int64_t *edge_ptr;
  const int32_t *rowptr;
  const int32_t *col;
  const int32_t *eorder;
  const uint8_t *nattr;
  const uint8_t *eattr;
  int32_t chunk_nodes; /* max over 64-graph groups [64i, 64i+64) of sum N_g; 0 = unknown (64*max_nodes assumed) */
  int32_t chunk_edges; /* same for sum E_g                                                                      */
  int32_t max_degree;  /* longest CSR row of the batch (upper bound accepted); 0 = unknown                       */
  int32_t reserved;    /* must be 0                                                                              */
  /* Byte-packed mirror of rowptr / col for batches of small graphs (max_edges <= 255, max_nodes <= 256), written
   * once per resident batch by gtok_csr_pack8 (same indexing as the int32 arrays); NULL = absent.  The
   * lane-per-graph SENT kernel stages its 64-graph chunks from these (a quarter of the index bytes).            */
  const uint8_t *rowptr8;
  const uint8_t *col8;
  /* Adjacency bit-matrix mirror for UNLABELLED batches of graphs with <= 256 nodes, written once per resident batch by
   * gtok_csr_adjbits; NULL = absent.  adj_rows[(node_ptr[g] + u) * adj_words + w] = word w of node u's row in the
   * symmetric closure of graph g's entries (self loops kept); adj_planes[(g * 8 + p) * adj_words + w] = bit p of the
   * degrees of nodes 64 w .. 64 w + 63; adj_max_degree = the largest degree (must be <= 255).  lane_order (optional):
   * the graphs in the order the lane-per-graph kernel deals them to lanes (64 per wave) - a permutation of 0..G-1
   * grouping walks of similar length; NULL = dataset order.  The SENT walk of such a batch reads one row per step.   */
  const uint64_t *adj_rows;
  const uint64_t *adj_planes;
  const int32_t *lane_order;
  int32_t adj_words;      /* 1, 2 or 4 = ceil(max_nodes / 64) rounded up to a power of two */
  int32_t adj_max_degree;
  /* A batch REORDERED for the lane-per-graph SENT kernel (optional; NULL / 0 = the batch is in dataset order).
   * graph_ids[s] = the dataset index of the graph stored at slot s: the row of the output slab, the entries of out_len
   * and `query`, and the RNG identity (graph_base + graph_ids[s]) follow it, so the result is that of the batch in
   * dataset order.  unit_ptr[num_units + 1] = the slots dealt to each wave (<= 64 per unit; chunk_nodes / chunk_edges
   * = the largest unit's sums).  Storing the graphs by descending walk length lets the 64 walks of a wave end together
   * (a wave lasts as long as its longest walk).  Only gtok_sent's lane-per-graph kernel accepts such a batch
   * (GTOK_E_INVAL from every other entry point and when another SENT kernel would be chosen).                      */
  const int32_t *graph_ids;
  const int32_t *unit_ptr;
  int32_t num_units;
  int32_t reserved2;
  /* Optional (NULL = derived from unit_ptr / node_ptr / edge_ptr by the kernel; ABI v4): one 32-byte record per unit of a
   * reordered batch, unit_info[8 u ..] = { unit_ptr[u], unit_ptr[u + 1], node_ptr[g0], node_ptr[gl], edge_ptr[g0] (low,
   * high word), edge_ptr[gl] (low, high) } - everything a wave needs to request its unit's chunk arrives in ONE scalar
   * load instead of a chain of two (a unit's staging is a chain of dependent HBM round trips: ~2 us each at launch start,
   * when every resident wave stages at once).                                                                      */
  const int32_t *unit_info;
} gtok_csr;

/* LUT layout for gtok_ibtt_zinc (int32 vocab ids; an absent token holds pad_id
 * exactly as TokenDataset's vocab.get(tok, vocab['<pad>']) would give):
 *   [0] <bos> [1] <eos> [2] <atom> [3] <bond> [4] <q> [5] regression [6] <p>
 *   [7..16]  atom symbols C N O F P S Cl Br I, then 'X'   (x outside 0..8)
 *   [17..21] 'unknown', single, double, triple, aromatic  (index = attr if 1..4 else 0)
 *   [22..22+n) decimal strings "0".."n-1"                  (node indices) */
#define GTOK_ZLUT_BOS 0
#define GTOK_ZLUT_EOS 1
#define GTOK_ZLUT_ATOM 2
#define GTOK_ZLUT_BOND 3
#define GTOK_ZLUT_Q 4
#define GTOK_ZLUT_REGRESSION 5
#define GTOK_ZLUT_P 6
#define GTOK_ZLUT_ATOM0 7
#define GTOK_ZLUT_BOND0 17
#define GTOK_ZLUT_NODE0 22

/* rowptr8[i] = (uint8_t)rowptr[i], col8[i] = (uint8_t)col[i] for the whole batch (sum N + G and sum E elements).
 * GTOK_E_TOO_LARGE when max_edges > 255 or max_nodes > 256 (values would not fit).  No reference counterpart: a
 * data-layout step, done once when a batch becomes resident, like the CSR build itself.                       */
int gtok_csr_pack8(const gtok_csr *g, int64_t num_rowptr, int64_t num_col, uint8_t *rowptr8, uint8_t *col8, void *stream);

/* What GTOK_CSR_SIMPLE_SYMMETRIC claims, verified on the device (ABI v5), and the batch maxima a caller may not know:
 * info = int32[8] in device memory (zeroed by the call), after the stream has run it holds
 *   [0] violations: 0 <=> every graph is simple and stored in both directions (no self loop, no (u,v) listed twice, (v,u)
 *       listed whenever (u,v) is) AND the arrays are well-formed (sizes >= 0, row pointers start at 0, never decrease and end
 *       at E_g, neighbour ids inside [0, N_g)) - only then may flags carry GTOK_CSR_SIMPLE_SYMMETRIC;
 *   [1] the longest row (max_degree)   [2] max N_g (max_nodes)   [3] max E_g (max_edges)   [4..7] 0.
 * Of *g only num_graphs and the five array pointers are read (col may be NULL when max_edges == 0 is claimed: the entry
 * checks are then skipped); a batch in dataset order only (graph_ids / unit_ptr set: GTOK_E_INVAL).  Nothing is read through an
 * offset that has not been range-checked first, so a damaged batch yields violations, never a fault - PROVIDED node_ptr /
 * edge_ptr hold num_graphs + 1 entries and rowptr / col as many elements as those two say.                                */
int gtok_csr_check(const gtok_csr *g, int32_t *info, void *stream);

/* The reordered copy of a batch of small graphs (max_nodes <= 64, max_edges <= 255: GTOK_E_TOO_LARGE otherwise) that
 * gtok_sent's lane-per-graph kernel walks fastest, built on the device in a dozen small launches and no host round trip
 * (ABI v5).  Graphs are stored by descending (nodes + rows of length 1) - the expected walk length: a walk restarts once per
 * dead end - ties in dataset order, and cut greedily into units of <= 64 graphs whose node and entry sums fit `lds_budget`
 * bytes of LDS per wave (0 = 10224: 16 resident waves per CU share its 160 KB), the budget being split between the two in
 * the corpus' own proportion.  Outputs (device memory, caller-allocated, G = num_graphs, N / E = the batch's node / entry
 * totals):
 *   graph_ids int32[G]        dataset index of the graph at every stored slot
 *   node_ptr  int32[G + 1], edge_ptr int64[G + 1], rowptr int32[N + G], col int32[E], nattr / eattr uint8[N] / [E] (only
 *                             when the batch has them): the batch in stored order, same layout as gtok_csr
 *   rowptr8   uint8[N + G + 16], col8 uint8[E + 16]   (optional, NULL = skip) the byte mirror of the two (gtok_csr_pack8)
 *   unit_ptr  int32[G + 1]    first slot of every unit, num_units + 1 entries used
 *   unit_info int32[8 G]      one record per unit (gtok_csr.unit_info), 8 num_units entries used
 *   info      int32[8]        as gtok_csr_check ([0] only when check != 0, else 0) + [4] num_units, [5] chunk_nodes,
 *                             [6] chunk_edges (the largest unit's node and entry sums)
 * The caller reads `info` back once (the only host read of the whole preparation), fills a gtok_csr with the new arrays,
 * flags (GTOK_CSR_SIMPLE_SYMMETRIC if verified), max_degree = info[1], chunk_nodes / chunk_edges, graph_ids, unit_ptr,
 * num_units, unit_info, rowptr8 / col8, and passes that to gtok_sent.  `workspace`: gtok_csr_lane_sort_workspace(G) bytes of
 * device memory, 16-byte aligned, free again when the stream has run the call.                                            */
typedef struct gtok_csr_sorted {
  int32_t *graph_ids;
  int32_t *node_ptr;
  int64_t *edge_ptr;
  int32_t *rowptr;
  int32_t *col;
  uint8_t *nattr;
  uint8_t *eattr;
  uint8_t *rowptr8;
  uint8_t *col8;
  int32_t *unit_ptr;
  int32_t *unit_info;
  int32_t *info;
} gtok_csr_sorted;
int64_t gtok_csr_lane_sort_workspace(int32_t num_graphs);
int gtok_csr_lane_sort(const gtok_csr *g, int32_t lds_budget, int32_t check, const gtok_csr_sorted *out, void *workspace,
                       int64_t workspace_bytes, void *stream);

/* Adjacency bit-matrix mirror (see gtok_csr.adj_rows): rows = [sum N_g][words] uint64, planes = [G][8][words] uint64,
 * info = int32[1], zeroed by the caller, receives the largest closure degree.  words = 1, 2 or 4 with 64 * words >=
 * max_nodes (GTOK_E_TOO_LARGE otherwise).  Works on any edge list (one or both directions, duplicates, self loops).
 * A data-layout step like gtok_csr_pack8: done once when a batch becomes resident, not per epoch.                    */
int gtok_csr_adjbits(const gtok_csr *g, int32_t words, uint64_t *rows, uint64_t *planes, int32_t *info, void *stream);

/* IBTT molecular serialiser -> ids.
 * lut_len = 22 + number of node-index entries (must cover max_nodes).      */
int gtok_ibtt_zinc(const gtok_csr *g, const int32_t *lut, int32_t lut_len,
                   int32_t max_len, int32_t pad_id, int32_t *out_ids,
                   int32_t ld, int32_t *out_len, void *stream);

/* LUT layout for gtok_ibtt_synth:
 *   [0] <bos> [1] <e> [2] <n> [3] <q> [4] <p>  [5..5+n) "0".."n-1"           */
#define GTOK_SLUT_BOS 0
#define GTOK_SLUT_E 1
#define GTOK_SLUT_N 2
#define GTOK_SLUT_Q 3
#define GTOK_SLUT_P 4
#define GTOK_SLUT_NODE0 5

/* graph-token grammar "<bos> u v <e> ... <n> 0 1 .. N-1 <q> q0 [q1 q2] <p>":
 * edges in ORIGINAL order (needs eorder unless identity), query[g*4+0] =
 * number of query ids (0..3), query[g*4+1..3] = the vocab ids after <q>.     */
int gtok_ibtt_synth(const gtok_csr *g, const int32_t *lut, int32_t lut_len,
                    const int32_t *query, int32_t max_len, int32_t pad_id,
                    int32_t *out_ids, int32_t ld, int32_t *out_len,
                    void *stream);

/* Open-addressing vocab table for gtok_text_to_ids, built on the host by
 * the Python mirror (capacity a power of two): slot s holds key_off[s] (byte
 * offset of the token string in key_bytes, -1 = empty), key_len[s], id[s].
 * Hash = 32-bit FNV-1a of the token bytes; probe linearly.                   */
typedef struct gtok_vocab_table {
  int32_t capacity;
  int32_t pad_id;
  const int32_t *key_off;
  const int32_t *key_len;
  const int32_t *id;
  const uint8_t *key_bytes;
} gtok_vocab_table;

/* TokenDataset: texts are concatenated ASCII bytes, text g = bytes
 * [text_ptr[g], text_ptr[g+1]).  Splits on Python str.split() ASCII
 * whitespace, keeps tokens up to and including the first "<p>" when
 * strip_label != 0, maps through the table (miss -> pad_id), cuts at
 * max_len.                                                                   */
int gtok_text_to_ids(const uint8_t *bytes, const int64_t *text_ptr,
                     int32_t num_texts, const gtok_vocab_table *vocab,
                     int32_t strip_label, int32_t max_len, int32_t *out_ids,
                     int32_t ld, int32_t *out_len, void *stream);

#define GTOK_SENT_SOS 0
#define GTOK_SENT_RESET 1
#define GTOK_SENT_LADJ 2
#define GTOK_SENT_RADJ 3
#define GTOK_SENT_EOS 4
#define GTOK_SENT_PAD 5
#define GTOK_SENT_IDX_OFFSET 6

/* gtok_sent_params.flags */
#define GTOK_SENT_NO_PAD 1 /* rows are only guaranteed up to out_len[g] (rounded up to a multiple of 16 ids); the pad tails -
                              56 % of a ZINC slab - may be left unwritten.  For callers that read rows through out_len
                              anyway (gtok_collate pads per batch, as both reference collates do).                  */
#define GTOK_SENT_U16 2    /* out_ids points at uint16_t storage: a [rows, ld] slab of 16-bit ids (ld still counts ids).  Every
                              SENT id fits (gtok_sent checks the id space and pad_id against 65535): the walk kernels hold
                              tokens 16 bits each anyway, so the rows leave as they are - half the bytes of the int32 slab,
                              no unpacking.  The reference never materialises a padded corpus slab (it pads per batch,
                              trainer/train_agtt.py:276-302); readers: gtok_collate_packed / gtok_unpack_rows with
                              row_ptr == NULL (the slab read in place) and gtok_pack_rows_u16.  The int32 slab stays the
                              documented default.                                                                   */

typedef struct gtok_sent_params {
  int32_t max_num_nodes;  /* tokenizer.set_num_nodes()                      */
  int32_t labeled;        /* labeled_graph: emit node/edge type tokens      */
  int32_t num_node_types; /* set_num_node_and_edge_types()                  */
  int32_t num_edge_types;
  int32_t max_len;        /* max_length == truncation_length                */
  int32_t remap_zinc;     /* fuse TokenizedGraphDataset.remap_zinc_tokens   */
  int32_t pad_id;         /* slab fill (Graph2TrailTokenizer.pad = 5)       */
  int32_t flags;          /* GTOK_SENT_* (0 = the documented output convention) */
  uint64_t seed;          /* Philox key                                     */
  uint64_t epoch;         /* a new trail every epoch                        */
  int64_t graph_base;     /* global index of graph 0 (shard-invariant RNG)  */
  const int32_t *query;   /* NULL, or [G,2] (query_u, query_v): appends
                             idx_off+N, idx_off+u, idx_off+v after the trail */
  int32_t epoch_count;    /* K epochs in ONE launch (0 and 1 both mean one): epoch slice e = 0..K-1 holds the trails of
                             epoch `epoch + e` - out_ids is [K, G, ld], out_len [K, G], `query` is shared.  The trainer
                             re-tokenizes the same split every epoch (trainer/train_agtt.py:246-250, epoch loop :676-680)
                             and a trail is a pure function of (seed, epoch, graph_base + g): K epochs of a small split
                             fill the chip where one epoch cannot (ABI v4).                                        */
  int32_t reserved;       /* must be 0                                      */
} gtok_sent_params;

/* SENT trail walk -> ids.  out_len[g] = min(trail, max_len) (+3 if query).  With epoch_count = K > 1 the result is
 * that of K calls with epoch, epoch + 1, ... written to consecutive [G, ld] slices.                               */
int gtok_sent(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids,
              int32_t ld, int32_t *out_len, void *stream);

/* gtok_sent whose walk ALSO appends every row to a packed buffer (ABI v6): the payload of the compact all-gather and of the
 * per-epoch D2H copy without a second pass over the rows.  `packed` holds `capacity` ids of the slab's width (16-bit with
 * GTOK_SENT_U16, else 32-bit).  The library cuts it into R <= GTOK_PACK_REGIONS equal regions of (capacity / R rounded down to a
 * multiple of 8) ids, each with a fill mark of its own - one mark for all 64-graph units costs as much as the walk - and deals
 * the units to the regions so that they fill evenly (R = the largest power of two that leaves every region >= 64 (unit, epoch)
 * pairs; a ZINC-full epoch: 32 regions that end within 1 % of each other): size `capacity` as the expected total + a few
 * per cent.  state (int64 [GTOK_PACK_STATE_WORDS], ZEROED BY THE CALLER before every launch): [0] collects status bits (bit 1:
 * a unit's rows did not fit its region; they are skipped and their row_start is -1), [GTOK_PACK_STATE_FILL + GTOK_PACK_STATE_STRIDE * r]
 * is region r's fill mark in ids (it counts skipped units too).  row_start[e * G + g] (int64 [K * G]) = first id of that row in
 * `packed`: rows start on 16-byte boundaries and lie in the order in which the units finished, NOT in dataset order - readers go
 * through row_start (gtok_unpack_rows_at, gtok_collate_packed with row_ptr = row_start).  Row r holds min(out_len[r], ld) ids;
 * the slab itself is written as gtok_sent writes it (pass GTOK_SENT_NO_PAD when nobody reads its pad tails).  Needs ld % 8 == 0
 * (% 4 for 32-bit ids), 16-byte aligned out_ids / packed, and a batch that runs the lane-per-graph molecule kernel (else
 * GTOK_E_UNSUPPORTED, nothing is launched: call gtok_sent + gtok_pack_rows_scan).                                           */
#define GTOK_SENT_PACK_ONLY 4 /* gtok_sent_params.flags, gtok_sent_packed only: the caller reads the PACKED rows alone - out_ids is then staging
                                space of gtok_sent_pack_scratch_rows() rows x ld ids (contents unspecified afterwards) instead of a
                                [K, G, ld] slab: every resident wave reuses its 64 rows for all of its units, the row stores stay in
                                the caches and K epochs need 92 MB of staging on an MI355X instead of K slabs.  Implies
                                GTOK_SENT_NO_PAD.                                                                              */
#define GTOK_PACK_REGIONS 64
#define GTOK_PACK_STATE_FILL 16    /* the fill marks start one cache line behind the status word ... */
#define GTOK_PACK_STATE_STRIDE 16  /* ... and keep a cache line each                                   */
#define GTOK_PACK_STATE_WORDS (GTOK_PACK_STATE_FILL + GTOK_PACK_STATE_STRIDE * GTOK_PACK_REGIONS)
int gtok_sent_packed(const gtok_csr *g, const gtok_sent_params *p, int32_t *out_ids, int32_t ld, int32_t *out_len,
                     void *packed, int64_t capacity, int64_t *row_start, int64_t *state, void *stream);
/* rows of staging space a GTOK_SENT_PACK_ONLY launch on the stream's device needs (64 per wave the device can hold: 262,144 on an
 * MI355X), or a negative GTOK_E_* code                                                                                          */
int64_t gtok_sent_pack_scratch_rows(void *stream);

/* remap_zinc_tokens over the first len[g] ids of every row, in place or not. */
int gtok_remap_zinc(const int32_t *in_ids, int32_t *out_ids, int32_t ld,
                    const int32_t *len, int32_t num_rows, int32_t idx_offset,
                    int32_t node_idx_offset, int32_t edge_idx_offset,
                    void *stream);

/* Batch collate: rows index[b] of the [*, ld] slab -> X[B, out_ld] int64
 * (pad_id beyond the row length; columns >= the batch max stay pad) and
 * attn[B, out_ld] uint8 (bool).  batch_max[0] receives max length.          */
int gtok_collate(const int32_t *ids, int32_t ld, const int32_t *len,
                 const int64_t *index, int32_t batch, int32_t pad_id,
                 int64_t *out_x, uint8_t *out_attn, int32_t out_ld,
                 int32_t *batch_max, void *stream);

/* graph-token text -> edge list: parse_graph_from_text / parse_query_nodes_from_text /
 * parse_label_from_text (graph_token_dataset_autograph.py:14-113) and the num_nodes rule of
 * parse_graph_from_json (:116-158) for texts in the canonical form
 *   <bos> (INT INT <e>)* <n> INT* [<q> WORD [INT INT]] [<p> WORD] ...
 * Texts as for gtok_text_to_ids.  Two passes: with edge_ptr == NULL only the per-text results are
 * written (num_edges[g], num_nodes[g], query_nodes[2g..] = (u, v) or (-1, -1), label[g] = 1/0 for
 * yes/no, K-1 for lenK, INT32_MIN for none, status[g]); the caller prefix-sums num_edges into
 * edge_ptr[G+1] and calls again with src/dst to receive the edges in text order.  status 1 marks a
 * text that is not in that form (token out of place, integer of more than 9 digits, several
 * <n>/<q>/<p>, `len` + junk): its other outputs are undefined and the host parser must take it. */
/* Sizing pass for gtok_parse_graph_text without the parse: num_edges[g] = the number of places in text g where the three
 * bytes `<e>` stand (for a text in the canonical form: its number of `<e>` tokens = its number of edges; for any text: never
 * less than its `<e>` tokens, so a range sized by it holds whatever the parse writes).  A streaming kernel, against one full
 * parse per pass of gtok_parse_graph_text: prefix-sum it into edge_ptr and make ONE parsing call with src / dst; a text's
 * edges are its first num_edges[g] slots as that call reports them (squeeze the ranges where the two counts differ).        */
int gtok_count_edge_tokens(const uint8_t *bytes, const int64_t *text_ptr, int32_t num_texts, int32_t *num_edges,
                           void *stream);
int gtok_parse_graph_text(const uint8_t *bytes, const int64_t *text_ptr, int32_t num_texts,
                          const int64_t *edge_ptr, int32_t *src, int32_t *dst, int32_t *num_edges,
                          int32_t *num_nodes, int32_t *query_nodes, int32_t *label, int32_t *status,
                          void *stream);

/* ---- packed (ragged) rows -------------------------------------------------------------------------------------
 * The padded slab is the documented output of every tokenizer entry point; more than half of a ZINC slab is padding
 * and every id fits 16 bits.  The packed form holds row r's n_r = min(len[r], ld) ids contiguously, `elem_bytes`
 * (2 or 4) bytes each, from element row_ptr[r] of `packed`.
 *
 * gtok_row_offsets: row_ptr[0] = 0, row_ptr[r + 1] = row_ptr[r] + round_up(n_r, align) for r < num_rows (int64
 * [num_rows + 1], align a power of two: 8 keeps every row start 16-byte aligned at 2 bytes per id - the fast path of
 * pack / unpack - 1 packs tightly).  Three small launches, no workspace.
 * gtok_pack_rows: slab -> packed, a buffer of `capacity` elements.  status[0] (zeroed by the caller) gets bit 0 when
 * elem_bytes == 2 and an id does not fit 16 bits (pack again with elem_bytes == 4), bit 1 when capacity <
 * row_ptr[num_rows] (rows that do not fit are skipped, nothing is written out of bounds).
 * gtok_unpack_rows: packed -> [num_rows, ld] slab, pad_id behind every row.  segment_rows > 0: the packed buffer is
 * the concatenation of equally sized segments - what an all-gather of per-rank buffers of segment_stride elements
 * gives - rows [s * segment_rows, (s + 1) * segment_rows) live in segment s: row r starts at
 * s * segment_stride + row_ptr[r] - row_ptr[s * segment_rows], with row_ptr = gtok_row_offsets over the gathered
 * lengths (same align).
 * gtok_collate_packed: gtok_collate reading the packed form (unsegmented) instead of the slab.
 * row_ptr == NULL (gtok_unpack_rows with segment_rows == 0, gtok_collate_packed): the STRIDED form - row r starts at
 * element r * ld of `packed` - i.e. a GTOK_SENT_U16 slab (or any [rows, ld] slab of 16- / 32-bit ids) read in place.
 * gtok_unpack_rows never reads beyond a segment: with segment_rows > 0 a row whose ids would end past segment_stride
 * (a rank whose rows did not fit the caller-given capacity skipped them in gtok_pack_rows, status bit 1) is written as
 * all pad and, when `status` is not NULL, flagged there (bit 1) - gtok_unpack_rows_checked is gtok_unpack_rows with that
 * status word and the total number of elements `packed` holds (0 = unknown) as further bounds.
 * gtok_pack_rows_u16: gtok_pack_rows reading a GTOK_SENT_U16 slab; elem_bytes 2, 4 or 8 (8: int64 ids, the dtype the
 * reference's tensors have - the D2H copy behind TokenizedGraphDataset.__getitem__ needs no widening pass).           */
int gtok_row_offsets(const int32_t *len, int64_t num_rows, int32_t ld, int32_t align, int64_t *row_ptr, void *stream);
int gtok_pack_rows(const int32_t *ids, int32_t ld, const int32_t *len, int64_t num_rows, const int64_t *row_ptr,
                   int32_t elem_bytes, void *packed, int64_t capacity, int32_t *status, void *stream);
int gtok_unpack_rows(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                     int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int32_t pad_id,
                     int32_t *out_ids, int32_t ld, void *stream);
int gtok_collate_packed(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                        int32_t ld, const int64_t *index, int32_t batch, int32_t pad_id, int64_t *out_x,
                        uint8_t *out_attn, int32_t out_ld, void *stream);
int gtok_unpack_rows_checked(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                             int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int64_t packed_elems,
                             int32_t pad_id, int32_t *out_ids, int32_t ld, int32_t *status, void *stream);
int gtok_pack_rows_u16(const uint16_t *ids16, int32_t ld, const int32_t *len, int64_t num_rows, const int64_t *row_ptr,
                       int32_t elem_bytes, void *packed, int64_t capacity, int32_t *status, void *stream);
/* gtok_unpack_rows_checked into a slab of 16-BIT ids (ABI v5; what GTOK_SENT_U16 writes and gtok_collate_packed reads in
 * place): the re-padding pass that ends a compact all-gather of 16-bit rows writes half the bytes.  ids keep their low 16
 * bits; 0 <= pad_id <= 65535.                                                                                             */
int gtok_unpack_rows_u16(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len,
                         int64_t num_rows, int32_t segment_rows, int64_t segment_stride, int64_t packed_elems,
                         int32_t pad_id, uint16_t *out_ids16, int32_t ld, int32_t *status, void *stream);
/* gtok_unpack_rows_checked / _u16 for rows with EXPLICIT starts (ABI v6; what gtok_sent_packed writes): row r's ids start at
 * element (r / segment_rows) * segment_stride + row_start[r] of `packed` (segment_rows == 0: at row_start[r]); a negative start,
 * or a row that would end beyond its segment / packed_elems (> 0), comes out as all pad and is flagged in status (bit 1).
 * out_bytes: 4 = an int32 slab, 2 = a slab of 16-bit ids.                                                                  */
int gtok_unpack_rows_at(const void *packed, int32_t elem_bytes, const int64_t *row_start, const int32_t *len, int64_t num_rows,
                        int32_t segment_rows, int64_t segment_stride, int64_t packed_elems, int32_t pad_id, void *out_ids,
                        int32_t out_bytes, int32_t ld, int32_t *status, void *stream);
/* gtok_row_offsets + gtok_pack_rows (src_bytes 4: an int32 slab) / gtok_pack_rows_u16 (src_bytes 2: a GTOK_SENT_U16 slab) in
 * ONE pass (ABI v5): row_ptr (int64 [num_rows + 1], OUT) and the packed rows are written by the same kernel - tiles of 256
 * rows, each tile learning its start from the tiles before it (single-pass prefix sum, the tile's closing row_ptr slot is its
 * status word) - for callers that know a capacity before they know the sizes (the compact all-gather with a caller-given
 * bound, a per-epoch D2H staging buffer): the lengths are read once, the ids once, and two launches replace four.  Same
 * results, same status bits - except that status is SET by this call, not accumulated into (bit 1 = rows that did not
 * fit `capacity` were skipped).                                                                                              */
int gtok_pack_rows_scan(const void *ids, int32_t src_bytes, int32_t ld, const int32_t *len, int64_t num_rows, int32_t align,
                        int32_t elem_bytes, void *packed, int64_t capacity, int64_t *row_ptr, int32_t *status, void *stream);

/* gtok_collate_packed for ONE batch whose row list is on the HOST (ABI v5; the stock torch DataLoader hands a dataset its index
 * list as Python ints - trainer/train_agtt.py:599-607): host_index[batch] (rows of the slab / packed form, each in [0, num_rows):
 * checked on the host, GTOK_E_INVAL otherwise) travels in the kernel's arguments - no upload, nothing to synchronise - and the
 * labels are gathered by the same launch: out_y[b] = y[host_index[b]] for elements of y_bytes (4: float32 / int32, 8: int64)
 * bytes each, y == NULL = none.  num_rows must fit 31 bits.  Outputs as gtok_collate_packed.                                  */
int gtok_collate_batch(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len, int32_t ld,
                       const int64_t *host_index, int32_t batch, int64_t num_rows, int32_t pad_id, int64_t *out_x, uint8_t *out_attn,
                       int32_t out_ld, const void *y, int32_t y_bytes, void *out_y, void *stream);

/* A whole EPOCH's batches collated by one call (ABI v5; trainer/train_agtt.py:599-607 builds the loader, :276-302 is the collate
 * it replaces): the rows order[0 .. n) of the slab / packed rows (as gtok_collate_packed reads them) are cut into batches of
 * batch_size consecutive entries of `order` (the last one may be short).  gtok_collate_epoch_plan writes batch_lmax[b] = the
 * longest row of batch b (min(len, ld) each; int32 [nb], nb = ceil(n / batch_size)) and batch_off[b] = the element at which
 * batch b starts in the arenas, batch_off[nb] = their size (int64 [nb + 1]).  gtok_collate_epoch then fills out_x (int64) and
 * out_attn (bool bytes): batch b is the row-major [B_b, batch_lmax[b]] block at batch_off[b] of both - exactly what
 * gtok_collate_packed gives for index = order[b * batch_size ..] and out_ld = batch_lmax[b].  arena_elems = the elements the
 * arenas hold: an upper bound such as n * ld needs no read-back before the fill (a row that would end beyond it is skipped).
 * A batch then costs its consumer no launch, no allocation and no synchronisation: the host reads batch_lmax / batch_off once
 * per epoch and takes views.                                                                                                  */
int gtok_collate_epoch_plan(const int32_t *len, int32_t ld, const int64_t *order, int64_t n, int32_t batch_size,
                            int32_t *batch_lmax, int64_t *batch_off, void *stream);
int gtok_collate_epoch(const void *packed, int32_t elem_bytes, const int64_t *row_ptr, const int32_t *len, int32_t ld,
                       const int64_t *order, int64_t n, int32_t batch_size, int32_t pad_id, const int32_t *batch_lmax,
                       const int64_t *batch_off, int64_t *out_x, uint8_t *out_attn, int64_t arena_elems, void *stream);

/* Rows of ids -> TEXT: the strings ZINCTokenizationDataset.__getitem__ returns (zinc_dataset_indexbase.py:143-227: the
 * trainer builds its vocab from them, trainer/train_ibtt.py:229-235, :361-372) rendered for a whole split at once.  Row
 * r's text = the strings of its first take[r] ids (string t = tab_bytes[tab_ptr[t] .. tab_ptr[t+1]), t < num_strings; ids
 * outside the table render as empty strings) joined by single spaces, followed VERBATIM by the row's suffix bytes
 * suf_bytes[suf_ptr[r] .. suf_ptr[r+1]) (suf_ptr NULL = none; the label token and <eos>, which differ per molecule, travel
 * here).  Two passes like gtok_parse_graph_text: out_bytes == NULL writes text_len[r]; the caller prefix-sums it into
 * text_ptr[num_rows+1] and calls again with out_bytes (capacity text_ptr[num_rows]).                                       */
int gtok_ids_to_text(const int32_t *ids, int32_t ld, const int32_t *take, int64_t num_rows, const uint8_t *tab_bytes,
                     const int32_t *tab_ptr, int32_t num_strings, const uint8_t *suf_bytes, const int64_t *suf_ptr,
                     const int64_t *text_ptr, uint8_t *out_bytes, int64_t *text_len, void *stream);

/* The per-molecule TAIL of a ZINC text, for gtok_ids_to_text's suffixes: zinc_dataset_indexbase.py:186-195 (`... <p> val_X_XX
 * <eos>`, the label token being f"val_{label:.2f}" with '.' -> '_' and '-' -> 'neg', :192) and :217-221 (a text of more than
 * max_len tokens keeps its first max_len - 1 tokens and `<eos>`).  len[r] = the ids row r holds (gtok_ibtt_zinc's length: up
 * to and including `<p>`), y[r] = the molecule's float32 label.  With cut = len[r] + 2 > max_len: take[r] = cut ? max_len - 1 :
 * len[r]; suffix r = [" " if take[r] > 0] + (cut ? "" : label token + " ") + "<eos>" (at most 56 bytes).  The label is "%.2f" of
 * the float's exact value with ties to even, as Python formats the double that holds it: the sign bit alone decides `neg`
 * (-0.001 -> val_neg0_00), values of any magnitude print in full, nan / inf / -inf -> val_nan / val_inf / val_neginf.
 * Two passes like gtok_ids_to_text: suf_bytes == NULL writes take[r] and suf_len[r]; the caller prefix-sums suf_len into
 * suf_ptr[num_rows+1] and calls again with suf_bytes (capacity suf_ptr[num_rows]).  max_len >= 1.                              */
int gtok_zinc_text_tails(const float *y, const int32_t *len, int64_t num_rows, int32_t max_len, int32_t *take,
                         const int64_t *suf_ptr, uint8_t *suf_bytes, int64_t *suf_len, void *stream);

/* First position of `token` in every row of an int64 [rows, ld] batch (what gtok_collate
 * returns): pos[r] = the smallest i with x[r, i] == token, -1 if there is none.  This is the
 * per-sample `<q>` search of trainer/train_ibtt.py:88-103 and trainer/train_agtt.py:78-114 (the
 * query nodes sit at pos + 2 and pos + 3) as one launch.                                      */
int gtok_find_token(const int64_t *x, int32_t rows, int32_t ld, int64_t token, int32_t *pos,
                    void *stream);

/* Node-id token statistics of the graph-token texts this batch stands for
 * (`<bos> u v <e> ... <n> 0 .. N-1 <q> TASK [qu qv] <p> LABEL <eos>`,
 * graph_token_dataset_autograph.py / docs/synthetic_data.md:46-68) - the corpus
 * pass of build_vocab_from_texts (data_loader.py:451-463) without the texts:
 * for every id i < num_ids,
 *   count[i] += occurrences of the token str(i): edge endpoints (list order),
 *               the <n> list, the two query arguments (query_nodes[g] = (u, v),
 *               NULL or negative entries = none);
 *   first[i]  = min(first[i], (graph_base + g) << 32 | token position in the text
 *               of graph g) over those occurrences.
 * The caller initialises count to 0 and first to INT64_MAX; calls ACCUMULATE, so
 * shards, ranks and epochs can be summed / min-reduced.  num_ids <= 1024.      */
int gtok_vocab_stats_synth(const gtok_csr *g, const int32_t *query_nodes,
                           int64_t graph_base, int32_t num_ids, int64_t *count,
                           int64_t *first, void *stream);

/* The same corpus pass over ARBITRARY texts (any grammar: graph-token tasks, the ZINC strings whose unseen
 * tokens become the dynamic vocab of trainer/train_ibtt.py:361-372): every distinct whitespace-separated token
 * (str.split() rules) gets one slot of an open-addressing table of `capacity` slots (a power of two):
 *   key[s]   64-bit identity of the token (two 32-bit hash streams; 0 = empty slot)
 *   count[s] occurrences            first[s] base_offset + byte offset of the earliest occurrence in `bytes`
 *   len[s]   token length in bytes  -> the token string is bytes[first - base_offset : .. + len]
 * The caller zeroes key / count / len and sets first to INT64_MAX; calls ACCUMULATE (shards: pass the shard's
 * offset in the whole corpus as base_offset so that `first` keeps ordering occurrences corpus-wide).  status
 * (zeroed by the caller) gets bit 0 when the table overflowed: enlarge and repeat; bit 2 (value 4) when two DIFFERENT
 * tokens met in one slot (same 64-bit identity and length, different bytes: every merge compares the occurrence with
 * the slot's first occurrence byte for byte) - the table is then not a token table and must be discarded.  The byte
 * check reaches occurrences inside the blob of the call at hand only: when shards are accumulated by several calls,
 * a collision between tokens of two different shards goes unseen, and reading the strings back needs the whole corpus
 * (a slot's `first` may lie in another shard).  Counter.most_common order =
 * count descending, then `first` ascending.                                                                    */
int gtok_vocab_stats_text(const uint8_t *bytes, const int64_t *text_ptr, int32_t num_texts, int64_t base_offset,
                          int32_t capacity, uint64_t *key, int64_t *count, int64_t *first, int32_t *len,
                          int32_t *status, void *stream);

/* SENT decoder: un-remapped token rows (gtok_sent with remap_zinc = 0) -> graphs in visit-index space: node
 * k is the k-th node the trail visited.  Per row: num_nodes, num_edges, the edges in stream order - edge_a =
 * the node the edge was written from, edge_b = the other end, edge_type = its type token minus the edge offset
 * (-1 unlabelled) - in [rows, edge_cap] arrays, node types in [rows, node_cap], and status: 0 complete (EOS
 * reached), 1 malformed, 2 a capacity exceeded (the row is still read to its end: num_nodes / num_edges are those of
 * the whole row, entries beyond a capacity are dropped - edge_cap = node_cap = 0 is a count-only pass), 3 well-formed but
 * cut before EOS (a row truncated at max_len).  The reverse of the spec in DESIGN.md section 5; with the trail's visit order it gives back
 * the input graph exactly (oracle_sent_roundtrip checks that for every row).                                      */
int gtok_sent_decode(const int32_t *ids, int32_t ld, const int32_t *len, int32_t num_rows,
                     int32_t max_num_nodes, int32_t labeled, int32_t num_node_types,
                     int32_t *num_nodes, int32_t *num_edges, int32_t *edge_a, int32_t *edge_b,
                     int32_t *edge_type, int32_t edge_cap, int32_t *node_type, int32_t node_cap,
                     int32_t *status, void *stream);

/* Which SENT kernel gtok_sent() will run for this batch ("sent_lane_kernel": lane per graph over the CSR, needs
 * GTOK_CSR_SIMPLE_SYMMETRIC, <= 64 nodes and a large batch; "sent_blane_kernel<W=..>": lane per graph over the
 * adjacency bit-matrix mirror, unlabelled, <= 256 nodes; "sent_reg_kernel": wave per graph, <= 64 nodes;
 * "sent_lds_kernel<W=..>": wave per graph, up to 512 nodes).  All emit the same tokens.                      */
const char *gtok_sent_kernel_name(const gtok_csr *g, const gtok_sent_params *p);
/* Same for gtok_ibtt_zinc(): "ibtt_zinc_quad_kernel" (8 or 16 lanes per molecule: GTOK_CSR_SIMPLE_SYMMETRIC
 * batches in list order - the default for ZINC), "ibtt_zinc_lane_kernel" (lane per graph, GTOK_IBTT_KERNEL=lane
 * only) or "ibtt_zinc_kernel" (wave per graph: any batch).  All emit the same tokens.                          */
const char *gtok_ibtt_zinc_kernel_name(const gtok_csr *g);

/* ABI version (GTOK_ABI_VERSION of the header the library was built from: 2 since gtok_csr carries the optional
 * mirrors - a binding checks it before passing structs; 3 adds the packed-row entry points; 4: gtok_sent_params carries
 * epoch_count, GTOK_SENT_U16, the strided / checked packed-row readers; 5 adds gtok_csr_check / gtok_csr_lane_sort -
 * structs unchanged; 6 adds gtok_sent_packed / GTOK_SENT_PACK_ONLY / gtok_unpack_rows_at / GTOK_E_UNSUPPORTED - structs
 * unchanged) and build target string ("gfx950").                     */
#define GTOK_ABI_VERSION 6
int gtok_version(void);
const char *gtok_target(void);

#ifdef __cplusplus
}
#endif
#endif /* GTOK_H */
