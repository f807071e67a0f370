// Internal declarations shared by the libddx translation units (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <memory>
#include <vector>

#include "../../include/ddx.h"

namespace ddx {

constexpr int kWave = 64;  // gfx950 wavefront
// Rows per panel of the column-major mirror.
// LDS-staged operator products (default): a slice of the float32 operand lives in LDS while the stored entries
// stream by.  kLdsPanelRows = rows of the row-major sketch per slice of the A^T Y pass = rows per panel of the
// column-major mirror (784 x 40 floats = 123 KB of the 160 KB LDS, the rest stages stored entries).
// DDX_SPMM=gather selects the L2-gather kernels, whose panels (kGatherPanelRows) are sized for one XCD's 4 MB L2.
#ifndef DDX_LDS_PANEL_ROWS
#define DDX_LDS_PANEL_ROWS 784
#endif
constexpr int kLdsPanelRows = DDX_LDS_PANEL_ROWS;
constexpr int kGatherPanelRows = 4096;   // measured optimum for the gather kernels (8192: +19 %, 2048: +5 %)
// Tuning / diagnostic switches of a context, set through ddx_set_option (include/ddx.h lists the keys); the library never
// reads the environment.  Switches that produce wrong results (timing ablations) exist only in builds with -DDDX_ABLATION.
struct Options {
    bool spmm_lds = true;
    bool gather_f32 = true;
    int spmm_geom = 0;               // 0 auto, 1 pair, 2 quad
    bool trip_packed = true;
    int knn_xcd_chunk = 32;          // DDX_KNN_XCD_CHUNK=n (0 = launch order): n consecutive query blocks of the MFMA passes share an XCD at a time
    bool knn_fold = true;            // DDX_KNN_FOLD=0: compare against the per-query threshold instead of folding it into the operands
    int64_t knn_sample_tiles = 0;    // 0 = default rule
    int knn_cells = 0;               // cells of the emit pass's pruning structure (0 = default rule, 1 = first-component windows only)
    int knn_sample_every = 32;       // the bound pass's sample holds every n-th tile of the whole set (0: none)
    int knn_seg_steps = 0;           // steps of a block's tile list per emit work item (0 = default)
    int knn_emit_waves = 0;          // waves per emit block: 4 or 8 (0 = default)
    int knn_emit_rt = 2;             // query tiles per emit wave: 2 (32 queries), 4 (64 queries, two waves per block; 32-component embeddings)
    bool row_sums_sequential = false;
    bool knn_debug = false;
    bool pca_debug = false;          // progress of the block Lanczos solver on stderr
    int bitplane = 1;                // entries equal to 1 as bitmaps on the int8 matrix cores (k_bitplane.hip): 0 off, 1 when the matrix is large enough, 2 always
    int bp_digits = 4;               // 8-bit digits of the operand's fixed point in those products (4: 30 bits below the column maximum, 3: 22)
    bool bp_mx = false;              // those products on the MX matrix instruction (FP4 bitmap x six base-31 digits in FP6, exact: k_bp_product6) instead of int8 digits
    int bp_dbg_mode = 0;             // only honoured under DDX_ABLATION -- TIMING ONLY (wrong results): bits 1 / 2 / 4 take the digit copies / bitmap copies / matrix instructions out of the MX product kernel's loop
    int bp_dbg_sk = 0;               // only honoured under DDX_ABLATION -- TIMING ONLY (wrong results): the matrix-core product kernels stop after this many stages per chunk
    int bp_digits_early = 0;         // ... in the power iterations before the last one of the randomized PCA (0 = as bp_digits, the default; 3: 22 bits -- 4 % faster fits, but the whole-fit comparison with the float64 oracle loses labels; 2: experiments)
    int mirror_mode = 2;             // column-major mirror: 2 counting sort placed by LDS tiles, 1 (DDX_MIRROR=scatter) counting sort with scattered stores, 0 (DDX_MIRROR=sort) radix sort
    bool upload_packed = true;       // DDX_UPLOAD=plain: send the raw matrix as it is (8 bytes per entry) instead of packed
    bool upload_form16 = true;       // DDX_UPLOAD=packed32: column | count << 16 (4 bytes per entry) instead of column step | count << 8 (2 bytes)
    bool upload_wait = false;        // DDX_UPLOAD=packed / packed32: wait for the pinned staging buffer instead of sending the first matrix plain (tests)
    bool arena_guard = false;        // DDX_ARENA_GUARD=1: pattern-fill the pad behind every block, ddx_check_memory verifies it
    int knn_ablation = 0;            // only honoured under DDX_ABLATION
    int upload_debug = 0;            // 1: timings of the upload on stderr, 2: per chunk
    bool residual_packed = true;     // bit-plane mode: the sparse products read this iteration's entries from wave-ordered packed blocks (k_spmm_packed) instead of the CSR / mirror (k_spmm_lds)
    int residual_rows_own = 12;      // outputs per lane group of the packed A Q kernel (12: one round of workgroups, each operand slice staged once per CU -- 0.157 ms per launch at the headline; 6: 0.177)
    bool synthetic_derived = true;   // bit-plane route: a doublet's bitmap row and reduced entries from its parents' (k_bp_synth); false: from the merged row
    int fault = 0;                   // fault injection (tests): 1 = allow_dynamic_lds fails
    bool testing = false;            // option testing=1 (the test suite): unlocks `fault`
    bool hvg_fold = true;            // gene sums folded in while the packed matrix arrives (off: one pass after the upload)
    int host_wait = 0;               // how a host thread waits for its stream (ddx::wait_stream): 0 the runtime's hipStreamSynchronize (spins), 1 "block": an event is polled
                                     // with sleeps in between (a waiting thread costs a few per cent of a CPU instead of a whole one)
    bool set(const char* key, const char* value);
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int blk = -1;                    // block of the context's arena that backs the buffer
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Device memory of a context comes from a few large chunks (one hipMalloc each) that are carved up by a bump pointer:
// a fit allocates ~50 buffers and hipFree costs ~0.2 ms apiece and synchronises the whole device, which would stall the
// other contexts (streams) of the GPU in the middle of their iterations.  Blocks are returned in any order; the bump
// pointer falls back over whatever is free at the top of its chunk (the stages' temporaries are LIFO in practice), the
// rest is reclaimed when the context is destroyed -- hipFree is called only there.
struct Arena {
    struct Chunk { void* p; size_t cap, off; };
    struct Block { int chunk; size_t off, size; bool free; };
    std::vector<Chunk> chunks;
    std::vector<Block> blocks;
    size_t next_chunk = (size_t)256 << 20;     // size of the next chunk to request (ddx reserves a better guess at upload)
    size_t peak = 0;                            // largest sum of the chunks' bump pointers since the context was created (ddx_arena_peak)
};

// Bit-plane operator products (k_bitplane.hip): bitmaps of the entries equal to 1 (all rows of the augmented matrix, by rows
// and by columns), the reduced sparse structures of the other entries (views into ctx->bp_buf built once per fit; the synthetic
// rows' parts are rewritten every iteration, their mirror lives in ctx->bp_synth), per-product work space (ctx->bp_work)
struct BitPlanes {
    bool ready = false;              // geometry + the original rows' structures built for this fit's counts
    bool values = false;             // synthetic rows' structures, reduced values and row scales refreshed for this iteration's matrix
    int SKc = 0;                     // stages of 256 matrix columns
    int64_t Npad = 0;                // original rows padded to a multiple of 256: the synthetic rows' padded indices start here
    int64_t ntile_o = 0, ntile_s = 0, ntile_c = 0;       // 32-row tiles: originals (padded), synthetic rows (this iteration), columns
    int64_t cap_rows = 0;            // padded rows the bitmaps have room for
    int64_t SKr = 0, SKr_cap = 0, SKr_used = 0;          // stages of 256 padded rows: layout stride of the column bitmap (= capacity) / in use
    int64_t nrest_o = 0, nrest_s = 0, cap_rest = 0, cap_rest_s = 0, cap_srow = 0;
    int64_t want_rest_s = 0;         // room for the synthetic rows' reduced entries asked for by an iteration that ran out of it
    size_t buf_bytes = 0;            // bytes of ctx->bp_buf in use (what a follower context copies)
    void* bm_rows = nullptr;         // [(row tile * SKc + sk) * 64 + r * 2 + h] 16-byte words
    void* bm_cols = nullptr;         // [(column tile * SKr + skr) * 64 + c * 2 + h]
    int64_t* rest_indptr = nullptr;  // reduced CSR of ALL rows: [M + 1]; cols / value; rest_raw, rest_row: count and row of an ORIGINAL row's entry
    int32_t* rest_cols = nullptr;    // (the values of those follow from them and the iteration's table; the synthetic rows' are derived from
    int32_t* rest_row = nullptr;     //  their parents' bitmaps and reduced rows: k_bp_synth)
    float* rest_raw = nullptr;
    float* rest_x = nullptr;
    int64_t* restm_colptr = nullptr; // reduced column-major mirror of the original rows: [P_o * H + 1], rows / counts / value
    int32_t* restm_row = nullptr;
    float* restm_raw = nullptr;      // (its counts: the values follow from them and the iteration's table)
    float* restm_x = nullptr;
    int64_t* restm_s_colptr = nullptr;   // ... of the synthetic rows: [P_s * H + 1], rows / value
    int32_t* restm_s_row = nullptr;
    float* restm_s_x = nullptr;
    double* srow = nullptr;          // [M] s_i = x_i(1) - z
    void* qd = nullptr;              // operand digits
    double* cmax = nullptr;          // [2][64] column maxima of the operand: Q side, Y side
    const double* ymax_of = nullptr; // the row-side matrix whose maxima (of diag(s) Y) the sparse A Q kernel has just left in cmax[64..]
    const double* qmax_of = nullptr; // the column-side matrix whose (weighted) maxima the Cholesky-QR's right multiplication has just left in cmax[0..63]
    int nd_now = 0;                  // digits of the products being issued (0: opt.bp_digits); stage_pca lowers it for its early power iterations
    bool qmax_zeroed = false;        // cmax[0..63] are zeros (the last Y-side digit kernel cleared them): maxima may be collected into them
    double* part = nullptr;          // partial blocks of the A^T Y product, one per chunk of the rows
    // standard scaling (sc.pp.scale, dd.py:302-303) on this route: an entry equal to 1 becomes s_i / sd_j as long as it is not clipped, so the
    // bitmaps stay what they are and 1 / sd_j goes into the operand (A Q) / the epilogue (A^T Y).  Columns in which an entry equal to 1 could
    // reach the clip are DEMOTED for the fit: none of their entries are in the bitmaps, all of them sit in the reduced structures with their own
    // (clipped) values.  demote: the per-column flags on the host (empty: none), copied with the struct to follower contexts; the device copy
    // is ctx->bp_demote
    std::vector<uint8_t> demote;
    int64_t n_demoted = 0;
    bool demote_decided = false;     // the first scaling of the fit has chosen the columns (with a margin); later ones only verify
    bool scaled = false;             // the reduced values, zcol and colmean describe the scaled matrix (bp_scale); inv_sd is valid
    const double* inv_sd = nullptr;  // [H] 1 / sd_j of this iteration (view into ctx->colstat)
    double* rowop = nullptr;         // [cap_srow x 3] per-row operand of the scale statistics (s, x(1)^2, 1)
};

// What another context of the same GPU copies when it takes over a context's resident counts (ddx_clone_counts, dd.py:178-184's
// memoised inputs).  Published under ddx_ctx::view_mu when the counts have become resident (end of stage_upload_counts) and
// withdrawn when the context starts its next fit.  Contract: the device pointers stay valid, and the bytes a follower copies from
// them stay constant, for as long as the view is valid -- a buffer that has to grow while the fit runs is ABANDONED to the arena
// (ensure_keep; bp_refresh's rebuild), never released for reuse, and later changes of the context (a full mirror built on demand,
// rebuilt bit planes) do not touch the view.  So the source may be in the middle of its iterations while followers copy.
struct CloneView {
    bool valid = false;
    int64_t N = 0, nnz = 0;
    int32_t H = 0, panel_rows = 0, P_o = 0;
    bool counts_exact = false, mirror_o = false;
    const void *aug_indptr = nullptr, *aug_indices = nullptr, *aug_raw = nullptr, *lib32 = nullptr, *lib64 = nullptr;
    const void *csc_o_colptr = nullptr, *csc_o_row = nullptr, *csc_o_raw = nullptr;
    std::vector<int64_t> h_indptr;
    BitPlanes bp;                    // the per-fit structures as they stood at publication (pointers into bp_buf)
    const void* bp_buf = nullptr;
};

struct TimingRec {
    int64_t launches = 0;
    double total_ms = 0.0;
};

struct PendingEvent {
    int name_id;
    hipEvent_t start, stop;
};

}  // namespace ddx

struct ddx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    ddx::Options opt;
    ddx::Arena arena;
    std::map<const void*, int> lds_configured;    // kernels whose dynamic-LDS limit was raised on this context's device, and to how many bytes
    std::string err;
    int64_t dev_bytes = 0;
    bool arena_hint_forced = false;   // ddx_reserve_hint: the library's own size guesses are ignored

    // ---- raw (all genes) matrix, only used by the on-device prologue -------------------------
    int64_t rawN = 0;
    int32_t rawG = 0;
    int64_t raw_nnz = 0;
    ddx::DevBuf raw_indptr, raw_indices, raw_data;
    std::vector<int64_t> h_raw_indptr;
    hipStream_t copy_stream = nullptr;   // the packed chunks travel on their own stream
    hipEvent_t wait_ev = nullptr;        // the event ddx::wait_stream polls (option host_wait=block)
    ddx::DevBuf raw_packed;          // the packed matrix on the device (expanded chunk by chunk)
    // gene sums accumulated while the matrix arrives (ddx_upload_raw, 2-byte form): per gene the two running float32 sums of
    // dd.py:167-170 in row order, carried from chunk to chunk; valid_rows = rows folded in so far (-1: none / another matrix)
    ddx::DevBuf hvg_state, hvg_keys, hvg_vals, hvg_colptr;
    int64_t hvg_rows = -1;
    int32_t hvg_G = 0;

    // ---- HVG-restricted counts, resident for the whole fit -------------------------------------
    int64_t N = 0;
    int32_t H = 0;
    int64_t nnz = 0;                 // stored entries of the N x H counts
    int64_t nnz_aug = 0;             // ... of the augmented matrix (known after ddx_lognormalise)
    std::vector<int64_t> h_indptr;   // host copy of the row pointer (capacity planning)
    int upload_form = 0;             // how the last ddx_upload_raw sent the matrix: 0 plain arrays, 1 packed here, 2 the packed image of another context's upload
    bool have_counts = false;
    bool counts_exact = false;       // counts are small non-negative integers: row sums are exact in any order

    // augmented matrix (rows 0..N-1 = originals, N..M-1 = synthetic doublets), CSR
    int64_t S = 0, M = 0;
    int64_t cap_synth = 0;           // capacity (entries) reserved behind the originals
    ddx::DevBuf aug_indptr;          // int64 [M+1]
    ddx::DevBuf aug_indices;         // int32 [nnz + cap_synth]
    ddx::DevBuf aug_raw;             // float [..] counts / summed counts
    ddx::DevBuf aug_x;               // float [..] value handed to PCA (log-normalised, maybe scaled)
    ddx::DevBuf lib32;               // float  [M] library sizes (float32 sequential sums)
    ddx::DevBuf lib64;               // double [M] L1 norms (double sequential sums)
    ddx::DevBuf synth_counts;        // int32 [S+1] scratch (row counts -> scan)
    ddx::DevBuf parents;             // int64 [S*2]
    ddx::DevBuf pad_off;             // int64 [S+1] padded row offsets of the doublet fill
    std::vector<int64_t> h_pad_off;
    bool have_synth = false;

    // column-major mirror: originals (static structure) + synthetic part (rebuilt per iteration)
    ddx::DevBuf csc_o_colptr;        // int64 [H+1]
    ddx::DevBuf csc_o_row;           // int32 [nnz]
    ddx::DevBuf csc_o_raw;           // float [nnz]
    ddx::DevBuf csc_o_x;             // float [nnz]
    ddx::DevBuf csc_s_colptr;        // int64 [H+1]
    ddx::DevBuf csc_s_row;           // int32 [cap_synth]  (row ids already offset by N)
    ddx::DevBuf csc_s_raw;           // float [cap_synth]
    ddx::DevBuf csc_s_x;             // float [cap_synth]
    ddx::DevBuf sort_keys_in, sort_keys_out, sort_vals_in, sort_vals_out, sort_tmp, sort_rowid;
    // the mirror is ordered by (row panel, column): entries of column j inside panel p form the segment
    // colptr[p*H + j] .. colptr[p*H + j + 1].  A panel is kPanelRows consecutive rows of the augmented
    // matrix, so the rows gathered while a panel is processed stay L2-resident.
    int32_t panel_rows = 784;        // fixed when the counts are uploaded (kLdsPanelRows or kGatherPanelRows)
    ddx::DevBuf rowseg;              // int32 [M x (slices+1)]: offset in row i of the first entry whose column is >= slice*SR
    ddx::DevBuf rank_buf;            // sort scratch + results of stage_rankings
    const int32_t* rank_rows = nullptr;   // rows by stored entries, descending (views into rank_buf)
    const int32_t* rank_cols = nullptr;   // columns likewise
    int32_t P_o = 0;                 // panels covering the original rows [0, N)
    int32_t p_s0 = 0, P_s = 0;       // first panel touched by synthetic rows, number of such panels

    // normalisation state
    float pseudocount = 0.1f;
    float zvalue = 0.f;              // value of the unstored entries before scaling: log(pseudocount), or 0 for log1p
    bool have_lognorm = false;
    bool scaled = false;
    float scale_max = 0.f;           // max_value of the scaling in force (ddx_scale); mean / sd of its columns: colstat + 2H / + 3H
    ddx::DevBuf median;              // float [1] (+ scratch)
    ddx::DevBuf lib_sorted;          // float [M]
    ddx::DevBuf lognorm_tab;         // float [M x 16] log-normalised value of the counts 1..16 in every row (row-major pass)
    ddx::DevBuf zcol;                // float  [H] value of unstored entries per column
    ddx::DevBuf colmean;             // double [H] mean over rows of (x - zcol) (0 for unstored)
    ddx::DevBuf colstat;             // double [2H] scratch for scale
    ddx::DevBuf col_part;            // double [2 x (panels x H)] per-(panel, column) partial sums

    // PCA work space
    int32_t C = 0;
    ddx::DevBuf pcaA, pcaB, pcaSmall, pcaPartial, pcaVec, pcaPanel, pcaOp, pcaQ0, pcaBlk;
    int64_t q0_rows = 0;             // shape of the start matrix kept in pcaQ0
    int32_t q0_cols = 0;
    ddx::DevBuf emb32;               // float  [M*C]
    ddx::DevBuf emb64;               // double [M*C]
    ddx::DevBuf sing;                // double [C]
    int64_t embM = 0;
    bool have_emb = false;

    // kNN
    int32_t K = 0;
    bool knn_self = false;
    ddx::DevBuf knn_idx;             // int32 [M*K]
    ddx::DevBuf knn_dist;            // double [M*K]
    ddx::DevBuf knn_sorted;          // int32 [M*K] neighbour lists sorted by index
    ddx::DevBuf edge_w;              // double [M*K]
    ddx::DevBuf knn_cells;           // cells, interval tables and chunk lists of the emit pass (stage_knn)
    ddx::DevBuf bp_buf, bp_work;     // bit-plane products: per-fit structures / per-product work space
    ddx::DevBuf bp_demote;           // uint8 [H] device copy of bp.demote (zeros: no column demoted)
    bool rows_scaled = false;        // aug_x holds the SCALED values of this iteration (the bit-plane route scales its reduced structures only)
    ddx::DevBuf bp_ms_colptr, bp_ms_row, bp_ms_x;   // ... the synthetic rows' reduced mirror (rebuilt every iteration)
    ddx::DevBuf pk_ptr[2], pk_blocks[2];   // packed residual products (k_pca.hip: k_pack_residual): [A Q, A^T Y] block tables and blocks of this iteration
    bool pk_valid[2] = {false, false};
    int64_t pk_nblocks[2] = {0, 0};
    bool synth_rows = true;          // rows N..M of the row-major arrays hold the current doublets (the bit-plane route derives its structures from
                                     // the parents' and leaves them out: ensure_full_rows builds them when somebody asks)
    bool rows_x = true;              // aug_x holds this iteration's values (same)
    bool mirror_o = false;           // csc_o_colptr / _row / _raw hold the original rows' full mirror (bit-plane route: built when somebody asks)
    bool mirror_full = false;        // csc_s_* and csc_*_x hold this iteration's full mirror (the bit-plane route leaves it out: ensure_full_mirror)
    ddx::BitPlanes bp;
    ddx::CloneView view;             // what followers copy (see CloneView); guarded by view_mu
    std::mutex view_mu;
    const int32_t* knn_overflow = nullptr;   // device counter: queries whose candidate list overflowed (exact rescan)
    const int32_t* knn_ccount = nullptr;     // candidates listed per query (kNN point order) and that order (views into the kNN work space)
    const int32_t* knn_perm = nullptr;
    bool have_knn = false;
    const unsigned long long* knn_window_total = nullptr;   // device counter: (query block, candidate tile) pairs screened by the emit pass
    double knn_window_pairs = 0.0;                          // the same count without pruning

    // graph: symmetric CSR left on the device by ddx_build_graph (views into graph_buf)
    ddx::DevBuf graph_buf;
    int64_t g_nodes = -1, g_entries = 0;
    const int64_t* g_d_indptr = nullptr;
    const int32_t* g_d_cols = nullptr;
    const double* g_d_vals = nullptr;

    // coarse graph left on the device by ddx_coarsen_graph (views into lv_buf)
    ddx::DevBuf lv_buf, lv_pack;
    void* lv_host = nullptr;         // pinned host copy of the packed coarse graph (w | indptr | member | cols)
    size_t lv_host_cap = 0;
    bool lv_host_valid = false;
    int64_t c_nodes = -1, c_entries = 0;
    // what part C needs of the levels of part A: graph of level l (l = 0: the graph above) and member table V_l -> V_{l+1}
    static constexpr int kLvKeep = 2;
    int lv_levels = 0;
    int64_t lv_m2 = 0;                                   // 2m of the quantised graph (the same on every level)
    int32_t lv_maxdeg[kLvKeep] = {0, 0}, lv_nbig[kLvKeep] = {0, 0};
    int64_t rowseg_rows = -1;                // original rows whose slice segments (rowseg) are valid for this fit's matrix
    int rowseg_ns = 0, rowseg_SR = 0;
    bool bp_rowseg = false;                  // the original rows' segments were cut from the reduced (bit-plane) rows
    bool lv_narrow[kLvKeep] = {};            // the level's edge weights fit the 32-bit register sweep
    int64_t lv_n[kLvKeep] = {0, 0}, lv_E[kLvKeep] = {0, 0};
    const int64_t* lv_indptr[kLvKeep] = {nullptr, nullptr};
    const int32_t* lv_cols[kLvKeep] = {nullptr, nullptr};
    const double* lv_w[kLvKeep] = {nullptr, nullptr};
    const int32_t* lv_member[kLvKeep] = {nullptr, nullptr};
    const int32_t* c_d_member = nullptr;
    const int64_t* c_d_indptr = nullptr;
    const int32_t* c_d_cols = nullptr;
    const double* c_d_vals = nullptr;

    // timing
    bool timing = false;
    std::vector<std::string> t_names;
    std::map<std::string, int> t_index;
    std::vector<ddx::TimingRec> t_recs;
    std::vector<ddx::PendingEvent> t_pending;
    std::vector<hipEvent_t> t_free;          // recycled events (creating a pair per scope cost more than recording it)
    std::map<const void*, int> t_by_ptr;     // scope names are string literals: their address identifies them
    // reference event (ddx_timing_reference): scope intervals are kept relative to it.  Shared between the contexts of
    // one GPU that report on one clock; the last holder destroys the event.
    std::shared_ptr<ihipEvent_t> t_ref;
    std::vector<float> t_intervals;          // (begin, end) in ms after t_ref of every timed scope since the last reset
};

namespace ddx {

int set_err(ddx_ctx* ctx, int code, const char* fmt, ...);
// wait until everything queued on the context's stream has finished (every host wait of the library goes through here: option host_wait)
hipError_t wait_stream(ddx_ctx* ctx);
int ensure(ddx_ctx* ctx, DevBuf& b, size_t bytes);
void release(ddx_ctx* ctx, DevBuf& b);
void arena_hint(ddx_ctx* ctx, size_t bytes);      // expected total need: sizes the next chunk
void arena_destroy(ddx_ctx* ctx);
void context_reset(ddx_ctx* ctx);
// raise a kernel's dynamic shared-memory limit once per context (the attribute is per device: a process-wide flag would
// leave the second GPU of a multi-GPU process with the 64 KB default)
int allow_dynamic_lds(ddx_ctx* ctx, const void* kernel, int bytes);
constexpr size_t kArenaPad = 4096;     // slack behind every block (tolerates the padded tail reads of the kernels)
void timing_begin(ddx_ctx* ctx, const char* name);
void timing_end(ddx_ctx* ctx);
int timing_flush(ddx_ctx* ctx);

#define DDX_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ddx::set_err((ctx), DDX_E_HIP, "%s failed: %s (%s:%d)", #call,            \
                                hipGetErrorString(e__), __FILE__, __LINE__);                 \
    } while (0)

#define DDX_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != DDX_OK) return rc__; \
    } while (0)

// RAII-free scoped kernel timer: KTIME(ctx, "name") { launches... }
struct ScopedTimer {
    ddx_ctx* c;
    ScopedTimer(ddx_ctx* ctx, const char* name) : c(ctx) { timing_begin(c, name); }
    ~ScopedTimer() { timing_end(c); }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }


// ---- log-normalisation of one stored entry (shared by k_sparse.hip and k_bitplane.hip) -------------------------
// the reference's element transform, evaluated in the reference's rounding order:
//   normed = float32( v / (double)rowsum )      sklearn inplace_csr_row_normalize_l1 (rowsum==0: unchanged)
//   scaled = normed * median                     float32 multiply                    (dd.py:293)
//   x      = log(scaled + pc)  |  log1p(scaled)  float32 result                      (dd.py:295 / :297)
// The log itself is evaluated in float64 and rounded once, i.e. the correctly rounded float32 value.
__device__ __forceinline__ float lognorm_value(float v, double rowsum, float med, float pc, bool use_log1p) {
#pragma clang fp contract(off)
    const float normed = (rowsum == 0.0) ? v : (float)((double)v / rowsum);
    const float scaled = normed * med;          // plain operators: the pragma above forbids fusing
    if (use_log1p) return (float)log1p((double)scaled);
    const float shifted = scaled + pc;
    return (float)log((double)shifted);
}

constexpr int kLognormTab = 16;     // counts 1..16 of every row are evaluated once per iteration (lognorm_tab)

__device__ __forceinline__ int small_count(float v, int kmax) {       // 1..kmax for such an integer count, else 0
    const int iv = (int)v;
    return (v == (float)iv && iv >= 1 && iv <= kmax) ? iv : 0;
}


// ---- stage entry points implemented in the .hip files ------------------------------------------
int stage_upload_counts(ddx_ctx* ctx, int64_t N, int32_t H, const int64_t* indptr, const int32_t* indices,
                        const float* data, bool from_device);
int stage_clone_counts(ddx_ctx* ctx, ddx_ctx* src);
void publish_clone_view(ddx_ctx* ctx);
int stage_create_doublets(ddx_ctx* ctx, int64_t S, const int64_t* parents);
int stage_lognormalise(ddx_ctx* ctx, float pseudocount);
int stage_scale(ddx_ctx* ctx, float max_value);
int stage_dense_rows(ddx_ctx* ctx, int64_t row0, int64_t nrows, float* out_host);
int stage_pca(ddx_ctx* ctx, int32_t C, int32_t oversample, int32_t n_iter, const double* q0, int64_t q0_rows);
int stage_operator_apply(ddx_ctx* ctx, int32_t transpose, const double* X, int32_t n, double* out);
int stage_pca_block_lanczos(ddx_ctx* ctx, int32_t C, int32_t oversample, double tol, int32_t max_steps, const double* q0, int32_t* steps_out,
                            ddx_eigh_fn eigh, void* eigh_user);
int stage_knn(ddx_ctx* ctx, int32_t k, int32_t include_self);
int stage_knn_metric(ddx_ctx* ctx, int32_t k, int32_t include_self, int32_t metric);
int stage_knn_candidate_counts(ddx_ctx* ctx, int32_t* host_out);
int stage_build_graph(ddx_ctx* ctx, int32_t mode);
int stage_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_host, double* w_host);
void assemble_graph(int64_t M, int K, const int32_t* idx, const double* w, std::vector<int64_t>& ip,
                    std::vector<int32_t>& gi, std::vector<double>& gw);
int validate_csr(ddx_ctx* ctx, const int64_t* indptr, const int32_t* cols, const float* vals, int64_t N, int32_t G);
// rankings of the rows / columns by stored entries; the arrays given are the ones the sparse products will walk (bit-plane mode: the reduced ones)
int stage_rankings(ddx_ctx* ctx, const int64_t* indptr, const int64_t* cp_o, const int64_t* cp_s);
int stage_coarsen_graph(ddx_ctx* ctx, double gamma, int32_t sweeps, int32_t levels);
int stage_refine_communities(ddx_ctx* ctx, const int32_t* coarse_labels, double gamma, int32_t sweeps, int32_t* labels_out);
int stage_gene_variances(ddx_ctx* ctx, float* var_out);
// fold the entries of rows [row0, row1) of the raw matrix (already on the device) into the running gene sums; row0 must be the
// number of rows folded in so far (0 starts over).  max_entries: the largest number of entries one call will see.
int gene_sums_fold(ddx_ctx* ctx, int32_t G, int64_t n_rows, int64_t row0, int64_t row1, int64_t e0, int64_t e1, int64_t max_entries);
int stage_select_columns(ddx_ctx* ctx, const int64_t* cols, int32_t n_cols);
// bit-plane products (k_bitplane.hip)
int bp_build(ddx_ctx* ctx);
int ensure_full_mirror(ddx_ctx* ctx);
int ensure_full_rows(ddx_ctx* ctx);           // rows N..M of the row-major arrays and aug_x, when the lean iteration left them out
int bp_synth_libs(ddx_ctx* ctx);
int scan_counts(ddx_ctx* ctx, const int32_t* in, int64_t n, int64_t base, int64_t* out);   // out[i] = base + sum of in[0..i), i = 0..n
bool bp_lean(const ddx_ctx* ctx);              // this context's iterations derive the synthetic rows' structures from the parents'
int bp_reduced_mirrors(ddx_ctx* ctx);
int bp_originals_mirror(ddx_ctx* ctx);
int bp_colmean(ddx_ctx* ctx, const double* parts, int nparts);
int bp_clone(ddx_ctx* ctx, const CloneView& src);
bool bp_wanted_at_upload(const ddx_ctx* ctx);
int bp_refresh(ddx_ctx* ctx);
int bp_scale(ddx_ctx* ctx, float max_value);          // sc.pp.scale on the bit-plane structures (k_sparse.hip)
int bp_scale_sums(ddx_ctx* ctx, const double** parts, int* chunks);     // per column: sums of s_i, x_i(1)^2 and 1 over the bitmap's entries
int bp_rebuild_demoted(ddx_ctx* ctx, const std::vector<uint8_t>& want);
int bp_rows_product(ddx_ctx* ctx, const double* Q, int L, double* Y);
int bp_cols_product(ddx_ctx* ctx, const double* Y, int L, const double** part, int* chunks);

// host-side numerics
void jacobi_eigh(int n, double* a /* n*n row-major, destroyed */, double* evals, double* evecs);

}  // namespace ddx
