// libddx C-ABI: context life cycle, memory, timing and the thin extern "C" wrappers.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <mutex>
#include <new>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <thread>
#include <time.h>

#include "ddx_internal.h"

namespace ddx {

static thread_local std::string g_tls_err;

int set_err(ddx_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_tls_err = buf;
    return code;
}

// Every host wait of the library.  The runtime's hipStreamSynchronize spins: a waiting lane thread keeps a CPU busy (0.83 s of CPU time per
// 0.12 s fit with seven lanes; the pods these GPUs come in allow 16 CPUs' worth of time).  Round 5 asked the runtime to block instead
// (hipSetDeviceFlags(hipDeviceScheduleBlockingSync)) and a process that had run 120 fits then hung at exit: hipFree's SyncAllStreams waited on a
// completion nobody signalled any more (profiles/r06_host_wait_hang.txt: stack of the hung process) -- the flag was set on a device torch had
// already made active.  So the library waits by itself and leaves the device's flags alone: record an event, poll it -- a short spin for the
// common few-microsecond waits, then sleeps of ~20 us between polls.
hipError_t wait_stream(ddx_ctx* ctx) {
    if (ctx->opt.host_wait != 1) return hipStreamSynchronize(ctx->stream);
    if (!ctx->wait_ev) {
        const hipError_t e = hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming);
        if (e != hipSuccess) { ctx->wait_ev = nullptr; return hipStreamSynchronize(ctx->stream); }
    }
    hipError_t e = hipEventRecord(ctx->wait_ev, ctx->stream);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 64; ++i) {
        e = hipEventQuery(ctx->wait_ev);
        if (e != hipErrorNotReady) return e;
    }
    const struct timespec nap = {0, 20000};
    for (;;) {
        e = hipEventQuery(ctx->wait_ev);
        if (e != hipErrorNotReady) break;
        nanosleep(&nap, nullptr);
    }
    (void)hipGetLastError();                     // (hipErrorNotReady is not an error to report later)
    return e;
}

void release(ddx_ctx* ctx, DevBuf& b) {
    Arena& A = ctx->arena;
    if (b.p && b.blk >= 0 && b.blk < (int)A.blocks.size()) {
        A.blocks[b.blk].free = true;
        // fall back over free blocks at the top (a block is at the top of its chunk iff it ends at the chunk's bump pointer)
        while (!A.blocks.empty()) {
            Arena::Block& t = A.blocks.back();
            Arena::Chunk& c = A.chunks[t.chunk];
            if (!t.free || t.off + t.size != c.off) break;
            c.off = t.off;
            A.blocks.pop_back();
        }
    }
    b.p = nullptr;
    b.cap = 0;
    b.blk = -1;
}

int allow_dynamic_lds(ddx_ctx* ctx, const void* kernel, int bytes) {
    if (ctx->opt.fault == 1) return set_err(ctx, DDX_E_HIP, "dynamic LDS limit of %d bytes refused (fault injection)", bytes);
    auto it = ctx->lds_configured.find(kernel);
    if (it != ctx->lds_configured.end() && it->second >= bytes) return DDX_OK;      // (a later, larger request raises the limit again)
    DDX_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->lds_configured[kernel] = bytes;
    return DDX_OK;
}

void arena_hint(ddx_ctx* ctx, size_t bytes) {
    if (ctx->arena_hint_forced) return;                  // ddx_reserve_hint decides
    if (!ctx->arena.chunks.empty()) return;              // a context that holds its first chunk grows by quarters of it (a parked context's next fit
                                                         // must not take a second full-size chunk because one buffer came out larger)
    if (bytes > ctx->arena.next_chunk) ctx->arena.next_chunk = bytes;
}

// A context starts a new fit: forget every buffer and result of the previous one, keep the chunks (their memory is
// handed out again from the start).  Called by the entry points that make counts resident.
void context_reset(ddx_ctx* ctx) {
    (void)wait_stream(ctx);
    {
        std::lock_guard<std::mutex> lock(ctx->view_mu);       // the buffers the view points into are about to be handed out again
        ctx->view = ddx::CloneView();
    }
    ctx->hvg_rows = -1;
    DevBuf* bufs[] = {&ctx->raw_indptr, &ctx->raw_indices, &ctx->raw_data, &ctx->raw_packed, &ctx->hvg_state, &ctx->hvg_keys, &ctx->hvg_vals, &ctx->hvg_colptr, &ctx->aug_indptr, &ctx->aug_indices,
                      &ctx->aug_raw, &ctx->aug_x, &ctx->lib32, &ctx->lib64, &ctx->synth_counts, &ctx->parents, &ctx->pad_off,
                      &ctx->csc_o_colptr, &ctx->csc_o_row, &ctx->csc_o_raw, &ctx->csc_o_x, &ctx->csc_s_colptr,
                      &ctx->csc_s_row, &ctx->csc_s_raw, &ctx->csc_s_x, &ctx->sort_keys_in, &ctx->sort_keys_out,
                      &ctx->sort_vals_in, &ctx->sort_vals_out, &ctx->sort_tmp, &ctx->sort_rowid, &ctx->median, &ctx->lib_sorted, &ctx->lognorm_tab,
                      &ctx->zcol, &ctx->colmean, &ctx->colstat, &ctx->col_part, &ctx->pcaA, &ctx->pcaB, &ctx->pcaSmall,
                      &ctx->pcaPartial, &ctx->pcaVec, &ctx->pcaPanel, &ctx->pcaOp, &ctx->pcaQ0, &ctx->pcaBlk, &ctx->rowseg, &ctx->rank_buf, &ctx->lv_buf, &ctx->lv_pack, &ctx->graph_buf, &ctx->emb32, &ctx->emb64, &ctx->sing, &ctx->knn_idx,
                      &ctx->knn_dist, &ctx->knn_sorted, &ctx->edge_w, &ctx->knn_cells, &ctx->bp_buf, &ctx->bp_work, &ctx->bp_demote, &ctx->bp_ms_colptr, &ctx->bp_ms_row, &ctx->bp_ms_x, &ctx->pk_ptr[0], &ctx->pk_ptr[1], &ctx->pk_blocks[0], &ctx->pk_blocks[1]};
    for (DevBuf* b : bufs) { b->p = nullptr; b->cap = 0; b->blk = -1; }
    ctx->arena.blocks.clear();
    for (auto& c : ctx->arena.chunks) c.off = 0;
    ctx->rawN = 0; ctx->rawG = 0; ctx->raw_nnz = 0;
    ctx->N = 0; ctx->H = 0; ctx->nnz = 0; ctx->S = 0; ctx->M = 0; ctx->cap_synth = 0;
    ctx->have_counts = ctx->have_synth = ctx->have_lognorm = ctx->scaled = ctx->have_emb = ctx->have_knn = false;
    ctx->q0_rows = 0; ctx->q0_cols = 0;
    ctx->rank_rows = ctx->rank_cols = nullptr;
    ctx->knn_window_total = nullptr;
    ctx->knn_overflow = nullptr;
    ctx->knn_ccount = nullptr;
    ctx->knn_perm = nullptr;
    ctx->g_nodes = -1; ctx->g_entries = 0; ctx->g_d_indptr = nullptr; ctx->g_d_cols = nullptr; ctx->g_d_vals = nullptr;
    ctx->c_nodes = -1; ctx->c_entries = 0; ctx->c_d_member = nullptr; ctx->c_d_indptr = nullptr; ctx->c_d_cols = nullptr; ctx->c_d_vals = nullptr;
    ctx->lv_host_valid = false;
    ctx->rowseg_rows = -1;
    ctx->pk_valid[0] = ctx->pk_valid[1] = false;
    ctx->mirror_full = false;
    ctx->mirror_o = false;
    ctx->synth_rows = ctx->rows_x = true;
    ctx->rows_scaled = false;
    ctx->bp = ddx::BitPlanes();
}

void arena_destroy(ddx_ctx* ctx) {
    for (auto& c : ctx->arena.chunks) (void)hipFree(c.p);
    ctx->arena.chunks.clear();
    ctx->arena.blocks.clear();
    ctx->dev_bytes = 0;
}

int ensure(ddx_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return DDX_OK;
    // contents are never needed across a growth: all callers refill after ensure()
    if (b.p) release(ctx, b);
    Arena& A = ctx->arena;
    const size_t want = (((bytes < 256 ? 256 : bytes) + 255) & ~(size_t)255) + kArenaPad;
    int ci = -1;
    for (int i = (int)A.chunks.size() - 1; i >= 0; --i)
        if (A.chunks[i].off + want <= A.chunks[i].cap) { ci = i; break; }
    if (ci < 0) {
        size_t cap = A.next_chunk > want ? A.next_chunk : want;
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess && cap > want) {      // the guess was too greedy for what is left: take what is needed
            (void)hipGetLastError();              // (the failure is sticky: a later DDX_HIP(hipGetLastError()) would report it)
            cap = want;
            e = hipMalloc(&p, cap);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return set_err(ctx, DDX_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", cap, hipGetErrorString(e));
        }
        A.chunks.push_back({p, cap, 0});
        ctx->dev_bytes += (int64_t)cap;
        A.next_chunk = std::max<size_t>((size_t)256 << 20, cap / 4);   // later chunks: a quarter of the first guess
        ci = (int)A.chunks.size() - 1;
    }
    Arena::Chunk& c = A.chunks[ci];
    // a block can only fall back when it is the last one recorded: keep blocks in allocation order per top chunk
    b.p = static_cast<char*>(c.p) + c.off;
    b.cap = want - kArenaPad;
    b.blk = (int)A.blocks.size();
    A.blocks.push_back({ci, c.off, want, false});
    c.off += want;
    {
        size_t used = 0;
        for (const auto& ch : A.chunks) used += ch.off;
        if (used > A.peak) A.peak = used;
    }
    if (ctx->opt.arena_guard)
        (void)hipMemsetAsync(static_cast<char*>(b.p) + b.cap, 0xA5, kArenaPad, ctx->stream);
    return DDX_OK;
}

// One switchboard for every tuning / diagnostic choice (ddx_set_option).  The library never reads the environment.
// Returns false for an unknown key or a value the key does not take.
bool Options::set(const char* key, const char* value) {
    const std::string k = key ? key : "", v = value ? value : "";
    auto on = [&]() { return !(v.empty() || v == "0" || v == "off" || v == "false"); };
    auto num = [&](long long lo, long long hi, long long* out) {
        char* end = nullptr;
        const long long x = std::strtoll(v.c_str(), &end, 10);
        if (v.empty() || (end && *end) || x < lo || x > hi) return false;
        *out = x;
        return true;
    };
    long long x = 0;
    if (k == "defaults") { *this = Options(); return true; }
    if (k == "spmm") { if (v == "lds") spmm_lds = true; else if (v == "gather") spmm_lds = false; else return false; return true; }
    if (k == "pca_gather") { if (v == "f32") gather_f32 = true; else if (v == "f64") gather_f32 = false; else return false; return true; }
    if (k == "spmm_geom") { if (v == "auto") spmm_geom = 0; else if (v == "pair") spmm_geom = 1; else if (v == "quad") spmm_geom = 2; else return false; return true; }
    if (k == "spmm_trip") { if (v == "packed") trip_packed = true; else if (v == "f64") trip_packed = false; else return false; return true; }
    if (k == "host_wait") { if (v == "block") host_wait = 1; else if (v == "spin" || v == "auto") host_wait = 0; else return false; return true; }
    if (k == "bitplane") { if (v == "auto") bitplane = 1; else if (!num(0, 2, &x)) return false; else bitplane = (int)x; return true; }
    if (k == "residual") { if (v == "packed") residual_packed = true; else if (v == "plain") residual_packed = false; else return false; return true; }
    if (k == "residual_rows_own") { if (!num(6, 12, &x) || (x != 6 && x != 12)) return false; residual_rows_own = (int)x; return true; }
    if (k == "synthetic") { if (v == "derived") synthetic_derived = true; else if (v == "merged") synthetic_derived = false; else return false; return true; }
    if (k == "bp_digits") { if (!num(3, 4, &x)) return false; bp_digits = (int)x; return true; }
    if (k == "bp_format") { if (v == "mx6") bp_mx = true; else if (v == "int8") bp_mx = false; else return false; return true; }
#ifdef DDX_ABLATION
    if (k == "bp_dbg_mode") { if (!num(0, 15, &x)) return false; bp_dbg_mode = (int)x; return true; }
    if (k == "bp_dbg_sk") { if (!num(0, 1 << 20, &x)) return false; bp_dbg_sk = (int)x; return true; }
#endif
    if (k == "bp_digits_early") { if (!num(0, 4, &x) || x == 1) return false; bp_digits_early = (int)x; return true; }
    if (k == "knn_fold") { knn_fold = on(); return true; }
    if (k == "knn_xcd_chunk") { if (!num(0, 4096, &x)) return false; knn_xcd_chunk = (int)x; return true; }
    if (k == "knn_sample_tiles") { if (!num(0, 1 << 24, &x)) return false; knn_sample_tiles = x; return true; }
    if (k == "knn_sample_every") { if (!num(0, 1 << 20, &x)) return false; knn_sample_every = (int)x; return true; }
    if (k == "knn_cells") { if (!num(0, 1024, &x)) return false; knn_cells = (int)x; return true; }
    if (k == "knn_seg_steps") { if (!num(0, 1 << 20, &x)) return false; knn_seg_steps = (int)x; return true; }
    if (k == "knn_emit_waves") { if (!num(0, 8, &x) || (x != 0 && x != 4 && x != 8)) return false; knn_emit_waves = (int)x; return true; }
    if (k == "knn_emit_rt") { if (!num(2, 4, &x) || x == 3) return false; knn_emit_rt = (int)x; return true; }
    if (k == "knn_debug") { knn_debug = on(); return true; }
    if (k == "testing") { testing = on(); return true; }
    if (k == "fault") { if (!testing || !num(0, 1, &x)) return false; fault = (int)x; return true; }      // (fault injection: only after testing=1)
    if (k == "pca_debug") { pca_debug = on(); return true; }
    if (k == "row_sums") { if (v == "auto") row_sums_sequential = false; else if (v == "sequential") row_sums_sequential = true; else return false; return true; }
    if (k == "mirror") { if (v == "tiles") mirror_mode = 2; else if (v == "scatter") mirror_mode = 1; else if (v == "sort") mirror_mode = 0; else return false; return true; }
    if (k == "upload") {
        if (v == "auto") { upload_packed = true; upload_form16 = true; upload_wait = false; }
        else if (v == "plain") { upload_packed = false; upload_form16 = true; upload_wait = false; }
        else if (v == "packed") { upload_packed = true; upload_form16 = true; upload_wait = true; }
        else if (v == "packed32") { upload_packed = true; upload_form16 = false; upload_wait = true; }
        else return false;
        return true;
    }
    if (k == "upload_debug") { if (!num(0, 2, &x)) return false; upload_debug = (int)x; return true; }
    if (k == "hvg_fold") { hvg_fold = on(); return true; }
    if (k == "arena_guard") { arena_guard = on(); return true; }
    if (k == "knn_ablation") {
#ifdef DDX_ABLATION
        if (!num(0, 255, &x)) return false;
        knn_ablation = (int)x;
        return true;
#else
        return false;                      // timing ablations give wrong results: only in -DDDX_ABLATION builds (profiles/tools)
#endif
    }
    return false;
}

static bool timing_event(ddx_ctx* ctx, hipEvent_t* ev) {
    if (!ctx->t_free.empty()) {
        *ev = ctx->t_free.back();
        ctx->t_free.pop_back();
        return true;
    }
    return hipEventCreate(ev) == hipSuccess;
}

void timing_begin(ddx_ctx* ctx, const char* name) {
    if (!ctx->timing) return;
    int id;
    auto pit = ctx->t_by_ptr.find(name);
    if (pit != ctx->t_by_ptr.end()) {
        id = pit->second;
    } else {
        auto it = ctx->t_index.find(name);
        if (it == ctx->t_index.end()) {
            id = (int)ctx->t_names.size();
            ctx->t_names.emplace_back(name);
            ctx->t_index[name] = id;
            ctx->t_recs.emplace_back();
        } else {
            id = it->second;
        }
        ctx->t_by_ptr[name] = id;
    }
    PendingEvent ev;
    ev.name_id = id;
    if (!timing_event(ctx, &ev.start) || !timing_event(ctx, &ev.stop)) return;
    (void)hipEventRecord(ev.start, ctx->stream);
    ctx->t_pending.push_back(ev);
}

void timing_end(ddx_ctx* ctx) {
    if (!ctx->timing || ctx->t_pending.empty()) return;
    (void)hipEventRecord(ctx->t_pending.back().stop, ctx->stream);
    if (ctx->t_pending.size() >= 2048) (void)timing_flush(ctx);
}

int timing_flush(ddx_ctx* ctx) {
    if (ctx->t_pending.empty()) return DDX_OK;
    DDX_HIP(ctx, wait_stream(ctx));
    for (auto& ev : ctx->t_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
            ctx->t_recs[ev.name_id].launches += 1;
            ctx->t_recs[ev.name_id].total_ms += ms;
            float t0 = 0.f;
            if (ctx->t_ref && hipEventElapsedTime(&t0, ctx->t_ref.get(), ev.start) == hipSuccess) {
                ctx->t_intervals.push_back(t0);
                ctx->t_intervals.push_back(t0 + ms);
            }
        }
        ctx->t_free.push_back(ev.start);
        ctx->t_free.push_back(ev.stop);
    }
    (void)hipGetLastError();                    // (a scope whose stop was never recorded must not surface as the next call's error)
    ctx->t_pending.clear();
    return DDX_OK;
}

}  // namespace ddx

using namespace ddx;

#define REQUIRE_CTX(ctx) \
    if (!(ctx)) return ddx::set_err(nullptr, DDX_E_ARG, "null context")
#define USE_DEVICE(ctx) DDX_HIP((ctx), hipSetDevice((ctx)->device))

extern "C" {

int ddx_abi_version(void) { return DDX_ABI_VERSION; }

const char* ddx_last_error(const ddx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_tls_err.c_str(); }

int ddx_device_count(int* count) {
    if (!count) return set_err(nullptr, DDX_E_ARG, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return set_err(nullptr, DDX_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return DDX_OK;
}

int ddx_create(int device, ddx_ctx** out) {
    if (!out) return set_err(nullptr, DDX_E_ARG, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(nullptr, DDX_E_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return set_err(nullptr, DDX_E_ARG, "device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return set_err(nullptr, DDX_E_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return set_err(nullptr, DDX_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, DDX_E_UNSUPPORTED, "libddx is built for gfx950 only; device %d is %s", device,
                       prop.gcnArchName);
    ddx_ctx* c = new (std::nothrow) ddx_ctx();
    if (!c) return set_err(nullptr, DDX_E_NOMEM, "out of host memory");
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return set_err(nullptr, DDX_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return DDX_OK;
}

int ddx_destroy(ddx_ctx* ctx) {
    if (!ctx) return DDX_OK;
    const bool dbg = ctx->opt.upload_debug > 0;
    auto step = [&](const char* what) { if (dbg) { fprintf(stderr, "[ddx_destroy %p] %s\n", (void*)ctx, what); fflush(stderr); } };
    step("set device");
    (void)hipSetDevice(ctx->device);
    step("stream synchronize");
    (void)wait_stream(ctx);
    step("timing flush");
    (void)timing_flush(ctx);
    for (hipEvent_t e : ctx->t_free) (void)hipEventDestroy(e);
    ctx->t_free.clear();
    ctx->t_ref.reset();
    step("context reset");
    context_reset(ctx);
    step("arena destroy (hipFree)");
    arena_destroy(ctx);
    step("host free");
    if (ctx->lv_host) (void)hipHostFree(ctx->lv_host);
    step("stream destroy");
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
    (void)hipStreamDestroy(ctx->stream);
    step("done");
    delete ctx;
    return DDX_OK;
}

int ddx_synchronize(ddx_ctx* ctx) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_HIP(ctx, wait_stream(ctx));
    return DDX_OK;
}

// name of the context buffer that block `blk` of the arena backs (diagnostics of ddx_check_memory)
static const char* buffer_name(const ddx_ctx* c, int blk) {
#define DDX_NAMED(b) {#b, &c->b}
    const struct { const char* name; const DevBuf* buf; } named[] = {
        DDX_NAMED(raw_indptr), DDX_NAMED(raw_indices), DDX_NAMED(raw_data), DDX_NAMED(raw_packed), DDX_NAMED(hvg_state), DDX_NAMED(hvg_keys), DDX_NAMED(hvg_vals),
        DDX_NAMED(hvg_colptr), DDX_NAMED(aug_indptr), DDX_NAMED(aug_indices), DDX_NAMED(aug_raw), DDX_NAMED(aug_x), DDX_NAMED(lib32), DDX_NAMED(lib64),
        DDX_NAMED(synth_counts), DDX_NAMED(parents), DDX_NAMED(pad_off), DDX_NAMED(csc_o_colptr), DDX_NAMED(csc_o_row), DDX_NAMED(csc_o_raw), DDX_NAMED(csc_o_x),
        DDX_NAMED(csc_s_colptr), DDX_NAMED(csc_s_row), DDX_NAMED(csc_s_raw), DDX_NAMED(csc_s_x), DDX_NAMED(sort_keys_in), DDX_NAMED(sort_keys_out), DDX_NAMED(sort_vals_in),
        DDX_NAMED(sort_vals_out), DDX_NAMED(sort_tmp), DDX_NAMED(sort_rowid), DDX_NAMED(rowseg), DDX_NAMED(rank_buf), DDX_NAMED(median), DDX_NAMED(lib_sorted),
        DDX_NAMED(lognorm_tab), DDX_NAMED(zcol), DDX_NAMED(colmean), DDX_NAMED(colstat), DDX_NAMED(col_part), DDX_NAMED(pcaA), DDX_NAMED(pcaB), DDX_NAMED(pcaSmall),
        DDX_NAMED(pcaPartial), DDX_NAMED(pcaVec), DDX_NAMED(pcaPanel), DDX_NAMED(pcaOp), DDX_NAMED(pcaQ0), DDX_NAMED(pcaBlk), DDX_NAMED(emb32), DDX_NAMED(emb64), DDX_NAMED(sing),
        DDX_NAMED(knn_idx), DDX_NAMED(knn_dist), DDX_NAMED(knn_sorted), DDX_NAMED(edge_w), DDX_NAMED(knn_cells), DDX_NAMED(bp_buf), DDX_NAMED(bp_work), DDX_NAMED(bp_demote),
        DDX_NAMED(bp_ms_colptr), DDX_NAMED(bp_ms_row), DDX_NAMED(bp_ms_x), DDX_NAMED(pk_ptr[0]), DDX_NAMED(pk_ptr[1]), DDX_NAMED(pk_blocks[0]), DDX_NAMED(pk_blocks[1]),
        DDX_NAMED(graph_buf), DDX_NAMED(lv_buf), DDX_NAMED(lv_pack)};
#undef DDX_NAMED
    for (const auto& e : named)
        if (e.buf->p && e.buf->blk == blk) return e.name;
    return "a buffer since released, abandoned or local to a stage";
}

int ddx_check_memory(ddx_ctx* ctx) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    if (!ctx->opt.arena_guard) return set_err(ctx, DDX_E_ARG, "the context was created without DDX_ARENA_GUARD=1");
    DDX_HIP(ctx, wait_stream(ctx));
    std::vector<unsigned char> h(kArenaPad);
    int n = 0;
    for (const auto& blk : ctx->arena.blocks) {
        if (blk.free) { ++n; continue; }
        const char* pad = static_cast<const char*>(ctx->arena.chunks[blk.chunk].p) + blk.off + blk.size - kArenaPad;
        DDX_HIP(ctx, hipMemcpy(h.data(), pad, kArenaPad, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < kArenaPad; ++i)
            if (h[i] != 0xA5)
                return set_err(ctx, DDX_E_NUMERIC, "buffer overflow: block %d = %s (%zu bytes at chunk %d + %zu) was written %zu bytes past its end",
                               n, buffer_name(ctx, n), blk.size - kArenaPad, blk.chunk, blk.off, i + 1);
        ++n;
    }
    return DDX_OK;
}

int ddx_reserve_hint(ddx_ctx* ctx, int64_t bytes) {
    REQUIRE_CTX(ctx);
    if (bytes < 0) return set_err(ctx, DDX_E_ARG, "negative size");
    ctx->arena.next_chunk = (size_t)bytes;
    ctx->arena_hint_forced = bytes > 0;
    return DDX_OK;
}

int ddx_trim(ddx_ctx* ctx, int64_t keep_bytes) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    context_reset(ctx);                                  // nothing of the last fit survives: every chunk is empty
    while (!ctx->arena.chunks.empty() && ctx->dev_bytes > keep_bytes) {
        Arena::Chunk c = ctx->arena.chunks.back();       // (the first chunk is the large one: it goes last)
        ctx->arena.chunks.pop_back();
        (void)hipFree(c.p);
        ctx->dev_bytes -= (int64_t)c.cap;
    }
    return DDX_OK;
}

int ddx_device_memory(ddx_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    size_t f = 0, t = 0;
    DDX_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return DDX_OK;
}

int ddx_device_bytes(const ddx_ctx* ctx, int64_t* bytes) {
    REQUIRE_CTX(ctx);
    if (!bytes) return DDX_E_ARG;
    *bytes = ctx->dev_bytes;
    return DDX_OK;
}

int ddx_arena_peak(const ddx_ctx* ctx, int64_t* bytes) {
    REQUIRE_CTX(ctx);
    if (!bytes) return DDX_E_ARG;
    *bytes = (int64_t)ctx->arena.peak;
    return DDX_OK;
}

// ---- prologue -------------------------------------------------------------------------------
static int check_csr(ddx_ctx* ctx, int64_t n, int32_t g, const int64_t* indptr, const int32_t* indices,
                     const float* data) {
    if (n <= 0 || g <= 0) return set_err(ctx, DDX_E_ARG, "empty matrix (%lld x %d)", (long long)n, g);
    if (!indptr) return set_err(ctx, DDX_E_ARG, "null indptr");
    if (indptr[0] != 0) return set_err(ctx, DDX_E_ARG, "indptr[0] must be 0");
    for (int64_t i = 0; i < n; ++i)
        if (indptr[i + 1] < indptr[i]) return set_err(ctx, DDX_E_ARG, "indptr not monotone at row %lld", (long long)i);
    if (indptr[n] > 0 && (!indices || !data)) return set_err(ctx, DDX_E_ARG, "null indices/data");
    return DDX_OK;
}

// ---- packed upload of the raw matrix ---------------------------------------------------------------------------------
// dd.py:149-160 hands fit() a host matrix; its 8 bytes per stored entry (int32 column, float32 count) are what the PCIe
// link carries at 57 GB/s -- 14 ms of a 210 ms fit.  Counts are small non-negative integers and columns ascend inside a
// row by small steps (30 000 genes / 950 entries per row): host threads pack an entry into 2 bytes,
//     step to the previous column of the row (1..255; the first entry steps from column -1) | count << 8 (0..255),
// straight into pinned memory, chunk by chunk, while the previous chunk is on the link.  An entry that does not fit
// (a longer step, a column order the validation will reject, a count that is fractional, negative, > 255, NaN, -0.0)
// gets the code 0 and travels whole in a side list of (position, column, value) triples, so ANY matrix arrives bit for
// bit; a matrix with more than one such entry in 32 is sent plain.  One kernel expands the rows once the last chunk is
// there (a wave per row: prefix sum of the steps, restarted at the listed entries).  DDX_UPLOAD=packed32 selects the
// first version of the scheme, column | count << 16 in 4 bytes, expanded chunk by chunk on arrival; there an entry that
// does not fit (count >= 65 536, fractional, ...) makes the whole call fall back to the plain copies.
__global__ void k_expand_packed(const uint32_t* __restrict__ packed, int64_t n, int32_t* __restrict__ idx, float* __restrict__ val) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t p = packed[t];
    idx[t] = (int32_t)(p & 0xffffu);
    val[t] = (float)(p >> 16);
}

// 2-byte form -> (column, value) arrays, one wave per row.  esc_pos ascends; an entry with code 0 is looked up there.
__global__ void __launch_bounds__(256) k_expand_packed16(const uint16_t* __restrict__ code, const int64_t* __restrict__ indptr, int64_t row0, int64_t n_rows,
                                                         const int32_t* __restrict__ esc_pos, const int32_t* __restrict__ esc_col,
                                                         const float* __restrict__ esc_val, int32_t n_esc,
                                                         int32_t* __restrict__ idx, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // rows row0 .. n_rows - 1
    if (row >= n_rows) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    int32_t carry = -1;                                   // column of the entry before the block
    for (int64_t base = b; base < e; base += 64) {
        const int64_t i = base + lane;
        const bool in = i < e;
        const uint32_t c = in ? code[i] : 256u;          // (a lane past the end: count 1, step 0 -> adds nothing, listed nowhere)
        const bool esc = in && (c & 255u) == 0u;
        int32_t abs_col = 0;
        float v = (float)(c >> 8);
        if (esc) {
            int lo = 0, hi = n_esc;                      // first listed position >= i (it is there)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (esc_pos[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
            }
            abs_col = esc_col[lo];
            v = esc_val[lo];
        }
        int32_t s = esc ? 0 : (int32_t)(c & 255u);       // inclusive prefix sum of the steps
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t t = __shfl_up(s, off, 64);
            if (lane >= off) s += t;
        }
        const unsigned long long listed = __ballot(esc);
        const unsigned long long upto = listed & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int r = upto ? 63 - __clzll((long long)upto) : 0;
        const int32_t col_r = __shfl(abs_col, r, 64), s_r = __shfl(s, r, 64);
        const int32_t col = upto ? col_r + (s - s_r) : carry + s;
        if (in) { idx[i] = col; val[i] = v; }
        carry = __shfl(col, 63, 64);                      // (lanes past the end repeat the last column: step 0)
    }
}

// host threads of the packed upload: created once per process, parked between calls (spawning 48 threads costs ~3 ms)
namespace {
class WorkerPool {
  public:
    explicit WorkerPool(int n) : n_(n) {
        for (int w = 0; w < n; ++w) threads_.emplace_back([this, w] { loop(w); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int size() const { return n_; }
    // runs job(w) on every worker; returns when all have finished
    void run(const std::function<void(int)>& job) {
        std::unique_lock<std::mutex> l(m_);
        job_ = &job; left_ = n_; ++epoch_;
        cv_.notify_all();
        done_.wait(l, [this] { return left_ == 0; });
        job_ = nullptr;
    }
    // the same, but the caller keeps working and calls wait() later
    void start(const std::function<void(int)>& job) {
        { std::lock_guard<std::mutex> l(m_); job_ = &job; left_ = n_; ++epoch_; }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return left_ == 0; });
        job_ = nullptr;
    }
  private:
    void loop(int w) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
                job = job_;
            }
            if (job) (*job)(w);
            { std::lock_guard<std::mutex> l(m_); if (--left_ == 0) done_.notify_all(); }
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* job_ = nullptr;
    int left_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};
std::mutex g_pool_mutex;                     // one packed upload at a time per process (they share the host cores anyway)
// pinned staging of the packed upload, one per process.  Pinning 0.37 GB takes ~70 ms, five uploads' worth: the first
// call that needs a (larger) buffer starts the allocation on a helper thread and sends its matrix plain; later calls
// find the buffer ready.  Never freed (a few hundred MB of host memory for the life of the process).
std::atomic<int> g_pin_state{0};             // 0 none / too small, 1 being allocated, 2 ready
void* g_pin_buf = nullptr;
size_t g_pin_bytes = 0;
size_t g_pin_failed = 0;                     // smallest size hipHostMalloc has refused in this process (0: none): not retried
std::thread* g_pin_thread = nullptr;         // the helper that pins the buffer (joined before the next one starts and at exit); on the heap so that a
                                             // forked child, where the parent's thread does not exist, can abandon the object instead of destroying it
void pin_allocate(size_t need, int device) {
    (void)hipSetDevice(device);
    if (g_pin_buf) { (void)hipHostFree(g_pin_buf); g_pin_buf = nullptr; g_pin_bytes = 0; }
    void* p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocPortable) == hipSuccess) { g_pin_buf = p; g_pin_bytes = need; g_pin_state.store(2, std::memory_order_release); }
    else { (void)hipGetLastError(); g_pin_failed = need; g_pin_state.store(0, std::memory_order_release); }
}
void pin_join() { if (g_pin_thread && g_pin_thread->joinable()) g_pin_thread->join(); }
WorkerPool* g_pool = nullptr;
pid_t g_pool_pid = 0;
std::atomic<int> g_upload_threads{0};        // ddx_set_upload_threads (0: the default rule)
struct PackEsc { int32_t pos, col; float val; };
// The packed image is the same for every GPU: while one context packs (it holds g_pool_mutex), other contexts of the process that
// stage the SAME host arrays at that moment (one leader per GPU, classifier.py:_stage) attach to its job and send the chunks it
// finishes to their own device from the same pinned buffer, instead of falling back to the plain arrays.
struct PackShare {
    const int64_t* indptr; const int32_t* indices; const float* data;      // the job's identity
    int64_t n_cells, nnz; int32_t n_genes; bool f16;
    int64_t chunk, nchunks; int T;
    unsigned char* pin;
    std::atomic<int>* done;                                  // [nchunks] threads that have finished chunk k
    std::atomic<int>* bad;                                   // the matrix turned out not to be packable
    const std::vector<std::vector<PackEsc>>* listed;         // [chunk][thread] entries outside the 2-byte form
    pid_t pid;                                               // (a forked child must not attach to its parent's job)
    int readers = 0;                                         // attached contexts still copying (g_share_mutex)
};
std::timed_mutex g_share_mutex;              // (timed: a forked child may inherit it locked by a thread that does not exist there)
std::condition_variable_any g_share_cv;
PackShare* g_share = nullptr;                // the job other contexts may attach to (nullptr: none / closed)
// (called with g_pool_mutex held.  Threads do not survive fork(): a child process builds its own pool; the parent's
// object is abandoned there -- never joined, never freed.)
WorkerPool* upload_pool() {
    unsigned hw = std::thread::hardware_concurrency();
    // 16 threads: the packing is bound by the host's memory system, not by the number of threads (8 / 16 / 48 threads: 6.7 / 6.9 / 6.6 ms
    // per upload at the headline shape at the end of round 5), and 48 threads burn 0.24 s of CPU time per upload against 0.1 s --
    // which counts where the container's CPU allowance is a fraction of the CPUs it shows (profiles/tools/cpu_quota_check.py)
    unsigned n = std::max(4u, std::min(16u, hw ? hw / 2 : 8u));
    if (const int req = g_upload_threads.load()) n = (unsigned)req;
    if (g_pool && g_pool_pid == getpid() && g_pool->size() != (int)n) { delete g_pool; g_pool = nullptr; }   // resized on request
    if (!g_pool || g_pool_pid != getpid()) {
        g_pool = new WorkerPool((int)n);
        g_pool_pid = getpid();
    }
    return g_pool;
}
}  // namespace

// entries [a, b) of the matrix -> 2-byte codes; entries that do not fit are appended to `esc` (ascending positions).
// Rows are taken one at a time: the first entry of a row steps from column -1, the others from their predecessor, which
// leaves a branch-free inner loop the compiler turns into vector code.
#define DDX_PACK16_BODY                                                                                                          \
    int64_t row = std::upper_bound(indptr, indptr + n_rows + 1, a) - indptr - 1;                                                 \
    for (int64_t i = a; i < b;) {                                                                                                \
        while (indptr[row + 1] <= i) ++row;                                                                                      \
        const int64_t e = std::min(b, indptr[row + 1]);                                                                          \
        const int64_t s = i;                                                                                                     \
        uint32_t any = 0;                                                                                                        \
        if (i == indptr[row]) {                                                                                                  \
            any |= pack16_one((uint32_t)idx[i] + 1u, val[i], out + i);                                                           \
            ++i;                                                                                                                 \
        }                                                                                                                        \
        for (; i < e; ++i) any |= pack16_one((uint32_t)idx[i] - (uint32_t)idx[i - 1], val[i], out + i);                          \
        if (any)                                                                                                                 \
            for (int64_t t = s; t < e; ++t)                                                                                      \
                if (out[t] == 0) esc.push_back({(int32_t)t, idx[t], val[t]});                                                    \
    }

static inline uint32_t pack16_one(uint32_t step, float v, uint16_t* out) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    const bool range = (v >= 0.0f) & (v <= 255.0f);                       // (false for NaN)
    const int32_t c = (int32_t)(range ? v : 0.0f);
    const bool ok = range & ((float)c == v) & (bits != 0x80000000u) & (step - 1u < 255u);
    *out = ok ? (uint16_t)(step | ((uint32_t)c << 8)) : (uint16_t)0;
    return ok ? 0u : 1u;
}
static void pack16_generic(const int64_t* indptr, int64_t n_rows, const int32_t* idx, const float* val, int64_t a, int64_t b, uint16_t* out,
                           std::vector<PackEsc>& esc) {
    DDX_PACK16_BODY
}
__attribute__((target("avx2"))) static void pack16_avx2(const int64_t* indptr, int64_t n_rows, const int32_t* idx, const float* val, int64_t a,
                                                       int64_t b, uint16_t* out, std::vector<PackEsc>& esc) {
    DDX_PACK16_BODY
}

// ---- one packing per NODE: the packed image in POSIX shared memory (one process per GPU, torchrun) -----------------------------------
// With one process per GPU every rank used to pack its own copy of the same matrix: 0.1 s of CPU time each, on hosts that allow 16 CPUs'
// worth of time for 8 ranks (dd.py:149-160 has no counterpart: the reference is one process).  Local rank 0 now packs into a shared
// segment that every rank of the node registers as pinned memory once; the others follow its progress chunk by chunk -- codes from the
// segment over their own PCIe link, the listed entries from the segment's side arrays -- exactly as the contexts of ONE process attach to a
// PackShare.  Generations: a rank's g-th shareable upload is generation g everywhere (the ranks call fit() in lockstep); the owner reuses
// the buffer for generation g once every follower has reported generation g - 1 finished or abandoned.  Anything unexpected -- a follower
// that is late, another matrix, a matrix larger than the segment, a timeout -- ends in that rank packing for itself: slower, never wrong.
constexpr int kShmMaxChunks = 64, kShmMaxRanks = 64;
constexpr uint32_t kShmMagic = 0x64647836u;
struct ShmHeader {
    std::atomic<uint32_t> magic;
    uint64_t codes_bytes, esc_cap;                       // capacities: bytes of codes, listed entries
    std::atomic<uint64_t> generation;                    // the job described below (0: none yet)
    int64_t n_cells, nnz, chunk, nchunks;
    int32_t n_genes, f16;
    uint64_t fingerprint;
    std::atomic<int32_t> bad;                            // the matrix turned out not to be packable (or the owner failed)
    std::atomic<int64_t> esc_end[kShmMaxChunks];         // -1 until chunk k's codes and listed entries are in place; then the listed entries up to and including it
    std::atomic<uint64_t> follower_gen[kShmMaxRanks];    // last generation local rank r has finished with (or given up on)
};
struct ShmState {
    std::string name;
    int local_rank = 0, local_world = 1;
    uint64_t my_gen = 0;                                 // shareable uploads this process has made
    ShmHeader* hdr = nullptr;
    unsigned char* codes = nullptr;
    int32_t* esc_pos = nullptr; int32_t* esc_col = nullptr; float* esc_val = nullptr;
    size_t map_bytes = 0;
    bool registered = false, failed = false, owner_created = false;
};
ShmState g_shm;
std::mutex g_shm_mutex;

static void shm_unlink_at_exit() { if (g_shm.owner_created && !g_shm.name.empty()) (void)shm_unlink(g_shm.name.c_str()); }

static uint64_t upload_fingerprint(int64_t n_cells, int64_t nnz, const int64_t* indptr, const int32_t* indices, const float* data) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    const int64_t sr = std::max<int64_t>(1, n_cells / 2048), se = std::max<int64_t>(1, nnz / 4096);
    for (int64_t r = 0; r <= n_cells; r += sr) mix((uint64_t)indptr[r]);
    mix((uint64_t)indptr[n_cells]);
    for (int64_t e = 0; e < nnz; e += se) { uint32_t b; memcpy(&b, &data[e], 4); mix(((uint64_t)(uint32_t)indices[e] << 32) | b); }
    return h;
}

// maps (owner: creates) the segment; false: sharing is off for this process from now on
static bool shm_attach(int device, size_t want_codes, int64_t want_esc) {
    ShmState& S = g_shm;
    if (S.failed) return false;
    if (S.hdr) return true;
    const bool owner = S.local_rank == 0;
    int fd = -1;
    size_t bytes = 0;
    const size_t hdr_bytes = (sizeof(ShmHeader) + 4095) & ~(size_t)4095;
    if (owner) {
        (void)shm_unlink(S.name.c_str());
        fd = shm_open(S.name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { S.failed = true; return false; }
        const size_t codes = ((want_codes + want_codes / 4 + ((size_t)64 << 20)) + 4095) & ~(size_t)4095;
        const uint64_t esc = (uint64_t)(want_esc + want_esc / 4 + 4096);
        bytes = hdr_bytes + codes + 12 * (size_t)esc;
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); (void)shm_unlink(S.name.c_str()); S.failed = true; return false; }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { (void)shm_unlink(S.name.c_str()); S.failed = true; return false; }
        ShmHeader* h = new (m) ShmHeader();
        h->codes_bytes = codes; h->esc_cap = esc;
        h->generation.store(0);
        h->bad.store(0);
        for (auto& e : h->esc_end) e.store(-1);
        for (auto& f : h->follower_gen) f.store(0);
        S.hdr = h; S.map_bytes = bytes; S.owner_created = true;
        static bool hooked = false;
        if (!hooked) { hooked = true; std::atexit(shm_unlink_at_exit); }
        h->magic.store(kShmMagic, std::memory_order_release);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(S.name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size > hdr_bytes) { bytes = (size_t)st.st_size; break; }
                close(fd); fd = -1;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { S.failed = true; return false; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { S.failed = true; return false; }
        S.hdr = static_cast<ShmHeader*>(m); S.map_bytes = bytes;
        while (S.hdr->magic.load(std::memory_order_acquire) != kShmMagic) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { S.failed = true; S.hdr = nullptr; return false; }
            std::this_thread::yield();
        }
    }
    unsigned char* base = reinterpret_cast<unsigned char*>(S.hdr);
    S.codes = base + hdr_bytes;
    S.esc_pos = reinterpret_cast<int32_t*>(S.codes + S.hdr->codes_bytes);
    S.esc_col = S.esc_pos + S.hdr->esc_cap;
    S.esc_val = reinterpret_cast<float*>(S.esc_col + S.hdr->esc_cap);
    // pinned for THIS process's copies (the copies of a chunk would otherwise be staged through the runtime's own bounce buffers)
    (void)hipSetDevice(device);
    S.registered = hipHostRegister(S.codes, S.hdr->codes_bytes, hipHostRegisterPortable) == hipSuccess;
    if (!S.registered) (void)hipGetLastError();
    return true;
}

int ddx_set_upload_share_impl(const char* name, int32_t local_rank, int32_t local_world) {
    std::lock_guard<std::mutex> lock(g_shm_mutex);
    ShmState& S = g_shm;
    const std::string nm = (name && *name && local_world > 1) ? std::string("/ddx_") + name : std::string();
    if (nm == S.name && local_rank == S.local_rank && local_world == S.local_world) return DDX_OK;
    if (S.hdr) {
        if (S.registered) (void)hipHostUnregister(S.codes);
        (void)munmap(S.hdr, S.map_bytes);
        if (S.owner_created) (void)shm_unlink(S.name.c_str());
    }
    S = ShmState();
    if (nm.empty() || local_rank < 0 || local_rank >= local_world || local_world > kShmMaxRanks) return DDX_OK;
    S.name = nm; S.local_rank = local_rank; S.local_world = local_world;
    return DDX_OK;
}

// The consumer side of a packed upload: sends the chunks of job `sh` to `ctx` as the packing threads finish them and expands them
// there.  Run by the context that packs and by every context attached to its job.  DDX_OK / 1 (not packable) / DDX_E_HIP.
static int send_packed(ddx_ctx* ctx, const PackShare& sh, bool owner, double t_in, ShmState* pub = nullptr) {
    const bool f16 = sh.f16;
    const int64_t nnz = sh.nnz, n_cells = sh.n_cells, chunk = sh.chunk, nchunks = sh.nchunks;
    const int32_t n_genes = sh.n_genes;
    const int64_t* indptr = sh.indptr;
    const int T = sh.T;
    const size_t esz = f16 ? sizeof(uint16_t) : sizeof(uint32_t);
    const size_t need = esz * (size_t)nnz;
    const int64_t esc_cap = f16 ? nnz / 32 + 1 : 0;      // listed entries the device buffer has room for
    const size_t codes_bytes = (need + 255) & ~(size_t)255;
    unsigned char* pin = sh.pin;
    std::atomic<int>& bad = *sh.bad;
    const std::vector<std::vector<PackEsc>>& listed = *sh.listed;
    const bool dbg = ctx->opt.upload_debug > 0;
    auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<hipEvent_t> ev(nchunks, nullptr);
    int rc = DDX_OK;
    // 2-byte form: rows complete after every chunk, the largest batch of entries they bring, the listed entries so far
    std::vector<int64_t> rows_done_after((size_t)nchunks, 0);
    int64_t rows_done = 0, fold_max = 1;
    bool fold_ok = f16 && ctx->opt.hvg_fold;
    std::vector<int32_t> pos, col;                        // (alive and never reallocated until the stream has taken them: the synchronisation below)
    std::vector<float> val;
    unsigned char* side = ctx->raw_packed.as<unsigned char>() + codes_bytes;
    int32_t* d_pos = reinterpret_cast<int32_t*>(side);
    int32_t* d_col = d_pos + esc_cap;
    float* d_val = reinterpret_cast<float*>(d_col + esc_cap);
    if (f16) {
        pos.reserve((size_t)esc_cap + 1); col.reserve((size_t)esc_cap + 1); val.reserve((size_t)esc_cap + 1);
        int64_t prev = 0;
        for (int64_t k = 0; k < nchunks; ++k) {
            const int64_t c1 = std::min(nnz, (k + 1) * chunk);
            const int64_t r1 = (std::upper_bound(indptr, indptr + n_cells + 1, c1) - indptr) - 1;      // rows whose last entry is in by now
            rows_done_after[k] = r1;
            fold_max = std::max(fold_max, indptr[r1] - indptr[prev]);
            prev = r1;
        }
    }
    for (int64_t k = 0; k < nchunks && rc == DDX_OK; ++k) {
        while (sh.done[k].load(std::memory_order_acquire) < T && !bad.load()) std::this_thread::yield();
        if (bad.load()) { rc = 1; break; }
        const int64_t c0 = k * chunk, len = std::min(nnz, c0 + chunk) - c0;
        unsigned char* dev = ctx->raw_packed.as<unsigned char>() + esz * c0;
        if (ctx->opt.upload_debug > 1) fprintf(stderr, "[ddx upload] chunk %lld packed +%.2f ms\n", (long long)k, clk() - t_in);
        if (hipMemcpyAsync(dev, pin + esz * c0, esz * len, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
            hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[k], ctx->copy_stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, ev[k], 0) != hipSuccess) { rc = DDX_E_HIP; break; }
        if (!f16) {
            k_expand_packed<<<(unsigned)((len + 255) / 256), 256, 0, ctx->stream>>>(reinterpret_cast<const uint32_t*>(dev), len, ctx->raw_indices.as<int32_t>() + c0,
                                                                                    ctx->raw_data.as<float>() + c0);
            if (pub) pub->hdr->esc_end[k].store(0, std::memory_order_release);
            continue;
        }
        // 2-byte form: the entries this chunk lists go behind those of the earlier chunks (ascending positions), then the rows
        // that END in this chunk are expanded and -- while the next chunk is on the link -- folded into the running gene sums
        // of dd.py:167-170 (the stream has been told to wait for the chunk: the event above)
        const size_t before = pos.size();
        for (int w = 0; w < T; ++w)
            for (const PackEsc& x : listed[(size_t)k * T + w]) { pos.push_back(x.pos); col.push_back(x.col); val.push_back(x.val); }
        const size_t added = pos.size() - before;
        if ((int64_t)pos.size() > esc_cap) { rc = 1; break; }
        if (pub) {                                            // the other ranks of the node: this chunk's codes are in the segment, its listed entries follow
            if (pos.size() > pub->hdr->esc_cap) { rc = 1; break; }
            if (added) {
                memcpy(pub->esc_pos + before, pos.data() + before, 4 * added);
                memcpy(pub->esc_col + before, col.data() + before, 4 * added);
                memcpy(pub->esc_val + before, val.data() + before, 4 * added);
            }
            pub->hdr->esc_end[k].store((int64_t)pos.size(), std::memory_order_release);
        }
        if (added &&
            (hipMemcpyAsync(d_pos + before, pos.data() + before, 4 * added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
             hipMemcpyAsync(d_col + before, col.data() + before, 4 * added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
             hipMemcpyAsync(d_val + before, val.data() + before, 4 * added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) { rc = DDX_E_HIP; break; }
        const int64_t row1 = rows_done_after[k];
        if (row1 > rows_done) {
            k_expand_packed16<<<(unsigned)((row1 - rows_done + 3) / 4), 256, 0, ctx->stream>>>(ctx->raw_packed.as<uint16_t>(), ctx->raw_indptr.as<int64_t>(), rows_done, row1, d_pos,
                                                                                             d_col, d_val, (int32_t)pos.size(), ctx->raw_indices.as<int32_t>(),
                                                                                             ctx->raw_data.as<float>());
            if (fold_ok && gene_sums_fold(ctx, n_genes, n_cells, rows_done, row1, indptr[rows_done], indptr[row1], fold_max) != DDX_OK) fold_ok = false;
            rows_done = row1;
        }
    }
    if (rc != DDX_OK && owner) bad.store(1);               // (stops the packing threads; an attached context's own failure is its own)
    if (pub && (rc != DDX_OK || bad.load())) pub->hdr->bad.store(1, std::memory_order_release);
    const double t_issued = clk();
    if (ctx->opt.upload_debug > 1) {
        for (int64_t k = 0; k < nchunks; ++k)
            if (ev[k]) { (void)hipEventSynchronize(ev[k]); fprintf(stderr, "[ddx upload] chunk %lld landed +%.2f ms\n", (long long)k, clk() - t_in); }
    }
    if (rc == DDX_OK && bad.load()) rc = 1;
    if (rc == DDX_OK && f16 && dbg) fprintf(stderr, "[ddx upload] %lld of %lld entries listed\n", (long long)pos.size(), (long long)nnz);
    if (rc != DDX_OK || !fold_ok) ctx->hvg_rows = -1;
    (void)hipStreamSynchronize(ctx->copy_stream);
    const double t_copied = clk();
    (void)wait_stream(ctx);                                   // (the pinned buffer and the host lists are reused / freed)
    if (dbg)
        fprintf(stderr, "[ddx upload] %d-byte form%s, %d threads, %lld chunks: last copy issued +%.2f ms, copies done +%.2f, expanded +%.2f\n",
                (int)esz, owner ? "" : " (another context's packing)", T, (long long)nchunks, t_issued - t_in, t_copied - t_in, clk() - t_in);
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    return rc;
}

// A follower rank's upload from the node's segment (generation `gen`): DDX_OK, 1 (not available / not packable: the caller packs for
// itself or sends the matrix plain) or DDX_E_HIP.  Mirrors send_packed's consumer loop with the segment as the source.
static int recv_shared(ddx_ctx* ctx, ShmState& S, uint64_t gen, int64_t n_cells, int64_t nnz, int32_t n_genes, bool f16, const int64_t* indptr, uint64_t fp) {
    ShmHeader* h = S.hdr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    auto give_up = [&](int rc) { h->follower_gen[S.local_rank].store(gen, std::memory_order_release); return rc; };
    while (h->generation.load(std::memory_order_acquire) < gen) {
        if (now() - t0 > std::chrono::milliseconds(1500)) return give_up(1);
        std::this_thread::yield();
    }
    if (h->generation.load(std::memory_order_acquire) != gen || h->n_cells != n_cells || h->nnz != nnz || h->n_genes != n_genes || (h->f16 != 0) != f16 ||
        h->fingerprint != fp || h->nchunks > kShmMaxChunks)
        return give_up(1);
    const int64_t chunk = h->chunk, nchunks = h->nchunks;
    const size_t esz = f16 ? sizeof(uint16_t) : sizeof(uint32_t);
    const size_t codes_bytes = (esz * (size_t)nnz + 255) & ~(size_t)255;
    const int64_t esc_cap = f16 ? nnz / 32 + 1 : 0;
    unsigned char* side = ctx->raw_packed.as<unsigned char>() + codes_bytes;
    int32_t* d_pos = reinterpret_cast<int32_t*>(side);
    int32_t* d_col = d_pos + esc_cap;
    float* d_val = reinterpret_cast<float*>(d_col + esc_cap);
    std::vector<hipEvent_t> ev((size_t)nchunks, nullptr);
    bool fold_ok = f16 && ctx->opt.hvg_fold;
    int64_t rows_done = 0, fold_max = 1, prev_rows = 0, listed = 0;
    std::vector<int64_t> rows_done_after((size_t)nchunks, 0);
    if (f16)
        for (int64_t k = 0; k < nchunks; ++k) {
            const int64_t c1 = std::min(nnz, (k + 1) * chunk);
            const int64_t r1 = (std::upper_bound(indptr, indptr + n_cells + 1, c1) - indptr) - 1;
            rows_done_after[k] = r1;
            fold_max = std::max(fold_max, indptr[r1] - indptr[prev_rows]);
            prev_rows = r1;
        }
    int rc = DDX_OK;
    for (int64_t k = 0; k < nchunks && rc == DDX_OK; ++k) {
        int64_t end = -1;
        const auto tk = now();
        while ((end = h->esc_end[k].load(std::memory_order_acquire)) < 0 && !h->bad.load(std::memory_order_acquire)) {
            if (h->generation.load(std::memory_order_acquire) != gen || now() - tk > std::chrono::milliseconds(3000)) { rc = 1; break; }
            std::this_thread::yield();
        }
        if (rc != DDX_OK) break;
        if (end < 0 || h->bad.load()) { rc = 1; break; }
        const int64_t c0 = k * chunk, len = std::min(nnz, c0 + chunk) - c0;
        unsigned char* dev = ctx->raw_packed.as<unsigned char>() + esz * c0;
        if (hipMemcpyAsync(dev, S.codes + esz * c0, esz * len, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
            hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[k], ctx->copy_stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, ev[k], 0) != hipSuccess) { rc = DDX_E_HIP; break; }
        if (!f16) {
            k_expand_packed<<<(unsigned)((len + 255) / 256), 256, 0, ctx->stream>>>(reinterpret_cast<const uint32_t*>(dev), len, ctx->raw_indices.as<int32_t>() + c0,
                                                                                    ctx->raw_data.as<float>() + c0);
            continue;
        }
        if (end > esc_cap) { rc = 1; break; }
        const int64_t added = end - listed;
        if (added > 0 &&
            (hipMemcpyAsync(d_pos + listed, S.esc_pos + listed, 4 * (size_t)added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
             hipMemcpyAsync(d_col + listed, S.esc_col + listed, 4 * (size_t)added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
             hipMemcpyAsync(d_val + listed, S.esc_val + listed, 4 * (size_t)added, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) { rc = DDX_E_HIP; break; }
        listed = end;
        const int64_t row1 = rows_done_after[k];
        if (row1 > rows_done) {
            k_expand_packed16<<<(unsigned)((row1 - rows_done + 3) / 4), 256, 0, ctx->stream>>>(ctx->raw_packed.as<uint16_t>(), ctx->raw_indptr.as<int64_t>(), rows_done, row1, d_pos,
                                                                                             d_col, d_val, (int32_t)listed, ctx->raw_indices.as<int32_t>(),
                                                                                             ctx->raw_data.as<float>());
            if (fold_ok && gene_sums_fold(ctx, n_genes, n_cells, rows_done, row1, indptr[rows_done], indptr[row1], fold_max) != DDX_OK) fold_ok = false;
            rows_done = row1;
        }
    }
    if (rc != DDX_OK || !fold_ok) ctx->hvg_rows = -1;
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)wait_stream(ctx);                                   // (the segment may be rewritten once this rank has reported)
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (rc == DDX_OK && (h->bad.load() || h->generation.load() != gen)) rc = 1;
    if (ctx->opt.upload_debug > 0) fprintf(stderr, "[ddx upload] local rank %d from the node's segment (generation %llu): rc %d\n", S.local_rank, (unsigned long long)gen, rc);
    return give_up(rc);
}

// device buffer of the packed form: the codes, then (2-byte form) room for the listed entries: positions | columns | values
static int packed_device_buffer(ddx_ctx* ctx, bool f16, int64_t nnz) {
    const size_t need = (f16 ? sizeof(uint16_t) : sizeof(uint32_t)) * (size_t)nnz;
    const int64_t esc_cap = f16 ? nnz / 32 + 1 : 0;
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->copy_stream = nullptr; return 1; }
    return ensure(ctx, ctx->raw_packed, ((need + 255) & ~(size_t)255) + 12 * (size_t)esc_cap);
}

// DDX_OK: the matrix is on the device; 1: not applicable (caller sends it plain)
static int upload_packed(ddx_ctx* ctx, int64_t n_cells, int64_t nnz, int32_t n_genes, const int64_t* indptr, const int32_t* indices,
                         const float* data) {
    const bool f16 = ctx->opt.upload_form16;
    if ((!f16 && n_genes > 65536) || nnz < ((int64_t)1 << 20) || nnz > ((int64_t)384 << 20)) return 1;
    // 16 chunks: the first is on the link 0.6 ms after the call, the copies then follow each other on their own stream
    const int64_t chunk = std::max<int64_t>((nnz + 15) / 16, (int64_t)1 << 18);
    const int64_t nchunks = (nnz + chunk - 1) / chunk;
    const size_t esz = f16 ? sizeof(uint16_t) : sizeof(uint32_t);
    const size_t need = esz * (size_t)nnz;
    const int64_t esc_cap = f16 ? nnz / 32 + 1 : 0;      // listed entries the device buffer has room for
    auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    ctx->upload_form = 0;
    // Several ranks on this node (ddx_set_upload_share): this is generation `gen` of the node's segment.  A follower takes the image from
    // there; the owner (local rank 0) packs into it below.  Whatever goes wrong ends in this rank packing for itself.
    ShmState* shm = nullptr;
    uint64_t gen = 0, fp = 0;
    {
        std::lock_guard<std::mutex> lock(g_shm_mutex);
        if (!g_shm.name.empty() && !g_shm.failed && nchunks <= kShmMaxChunks) {
            gen = ++g_shm.my_gen;
            if (shm_attach(ctx->device, need, esc_cap)) shm = &g_shm;
        }
    }
    if (shm) fp = upload_fingerprint(n_cells, nnz, indptr, indices, data);
    if (shm && shm->local_rank != 0) {
        int rc = packed_device_buffer(ctx, f16, nnz);
        if (rc == DDX_OK) {
            (void)wait_stream(ctx);
            rc = recv_shared(ctx, *shm, gen, n_cells, nnz, n_genes, f16, indptr, fp);
        }
        if (rc == DDX_E_HIP) return set_err(ctx, DDX_E_HIP, "packed upload from the node's shared image failed");
        if (rc < 0) return rc;
        if (rc == DDX_OK) { ctx->upload_form = 2; return DDX_OK; }
        shm = nullptr;                                            // not available: this rank packs for itself
    }
    // the owner publishes generation `gen` on every path from here on -- as unusable (bad) unless it gets as far as packing into the segment
    struct ShmPublish {
        ShmState* s; uint64_t gen; bool done = false;
        ~ShmPublish() {
            if (s && !done) { s->hdr->bad.store(1); s->hdr->generation.store(gen, std::memory_order_release); }
        }
    } shm_pub{shm, gen};
    bool shm_serve = false;
    if (shm) {
        ShmHeader* h = shm->hdr;
        shm_serve = need <= h->codes_bytes && (uint64_t)esc_cap <= h->esc_cap;
        // every follower has finished with (or given up on) the previous generation: the buffer may be rewritten
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 1; r < shm->local_world && shm_serve; ++r)
            while (h->follower_gen[r].load(std::memory_order_acquire) + 1 < gen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000)) { shm_serve = false; break; }
                std::this_thread::yield();
            }
    }
    // one packing at a time per process.  A second context staging at the same moment (several GPUs driven by one process) attaches
    // to the running job when it is packing the same host arrays, and sends its copy plain, in parallel, when it is not.
    // (g_share_mutex is held from before the attempt on g_pool_mutex until the job is published: whoever fails to get the pool
    // then sees the job of whoever got it.)
    std::unique_lock<std::timed_mutex> share_lock(g_share_mutex, std::chrono::milliseconds(500));
    if (!share_lock.owns_lock()) return 1;
    std::unique_lock<std::mutex> pool_lock(g_pool_mutex, std::try_to_lock);
    if (!pool_lock.owns_lock()) {
        PackShare* sh = g_share;
        // (the row pointer is compared by content: the Python binding widens scipy's 32-bit indptr into a fresh array per call)
        if (!sh || sh->pid != getpid() || sh->indices != indices || sh->data != data || sh->n_cells != n_cells || sh->nnz != nnz ||
            sh->n_genes != n_genes || sh->f16 != f16 || sh->bad->load() ||
            (sh->indptr != indptr && memcmp(sh->indptr, indptr, sizeof(int64_t) * (size_t)(n_cells + 1)) != 0))
            return 1;
        ++sh->readers;
        share_lock.unlock();
        const double t_in = clk();
        int rc = packed_device_buffer(ctx, f16, nnz);
        if (rc == DDX_OK) {
            (void)wait_stream(ctx);       // the copies must not overtake whatever the main stream still does with the device buffers
            rc = send_packed(ctx, *sh, false, t_in);
        }
        share_lock.lock();
        --sh->readers;
        g_share_cv.notify_all();
        share_lock.unlock();
        if (rc == DDX_E_HIP) return set_err(ctx, DDX_E_HIP, "packed upload failed");
        if (rc < 0) return rc;
        if (rc != DDX_OK) release(ctx, ctx->raw_packed);
        else ctx->upload_form = 2;
        return rc;
    }
    if (!shm_serve) {
        static pid_t pin_pid = 0;
        if (pin_pid != getpid()) {                               // (a forked child starts over; the parent's helper thread does not exist here)
            pin_pid = getpid(); g_pin_state.store(0); g_pin_buf = nullptr; g_pin_bytes = 0; g_pin_failed = 0;
            g_pin_thread = nullptr;                              // (abandoned, never joined or freed: it belongs to the parent)
            g_share = nullptr;
        }
        const int st = g_pin_state.load(std::memory_order_acquire);
        if (st == 1) return 1;                                   // still being pinned
        if (st == 0 || g_pin_bytes < need) {
            const size_t want = need + need / 8;                 // (some room for the next, slightly larger matrix)
            if (g_pin_failed && want >= g_pin_failed) return 1;  // the host refused to pin this much before: stay plain
            pin_join();                                          // (a finished helper of an earlier, smaller request)
            g_pin_state.store(1);
            if (ctx->opt.upload_wait) pin_allocate(want, ctx->device);
            else {
                // a process that exits while the helper is still pinning must not run hipHostMalloc during runtime teardown
                static bool hooked = false;
                if (!hooked) { hooked = true; std::atexit(pin_join); }
                delete g_pin_thread;                             // (joined above)
                g_pin_thread = new std::thread(pin_allocate, want, ctx->device);
                return 1;
            }
            if (g_pin_state.load() != 2) return 1;
        }
    }
    {
        const int rc = packed_device_buffer(ctx, f16, nnz);
        if (rc != DDX_OK) return rc;
    }
    unsigned char* pin = shm_serve ? shm->codes : static_cast<unsigned char*>(g_pin_buf);
    WorkerPool* pool = upload_pool();
    const int T = pool->size();
    std::vector<std::atomic<int>> done(nchunks);
    for (auto& d : done) d.store(0);
    std::atomic<int> bad{0};
    std::atomic<int64_t> n_listed{0};
    std::vector<std::vector<PackEsc>> listed(f16 ? (size_t)nchunks * T : 0);     // [chunk][thread]: ascending positions in that order
    const bool avx2 = __builtin_cpu_supports("avx2");
    const std::function<void(int)> worker = [&](int w) {
        for (int64_t k = 0; k < nchunks; ++k) {
            if (bad.load(std::memory_order_relaxed)) return;
            const int64_t c0 = k * chunk, c1 = std::min(nnz, c0 + chunk), len = c1 - c0;
            const int64_t a = c0 + len * w / T, b = c0 + len * (w + 1) / T;
            if (f16) {
                std::vector<PackEsc>& mine = listed[(size_t)k * T + w];
                if (avx2) pack16_avx2(indptr, n_cells, indices, data, a, b, reinterpret_cast<uint16_t*>(pin), mine);
                else pack16_generic(indptr, n_cells, indices, data, a, b, reinterpret_cast<uint16_t*>(pin), mine);
                if (!mine.empty() && n_listed.fetch_add((int64_t)mine.size()) + (int64_t)mine.size() > esc_cap) { bad.store(1); return; }
            } else {
                uint32_t* out = reinterpret_cast<uint32_t*>(pin);
                bool ok = true;
                for (int64_t i = a; i < b; ++i) {
                    const float v = data[i];
                    const uint32_t iv = (uint32_t)(int32_t)v;                    // (garbage for NaN / huge values: caught by the comparison)
                    const uint32_t j = (uint32_t)indices[i];
                    ok = ok && (float)iv == v && iv < 65536u && j < 65536u && !(iv == 0u && std::signbit(v));   // (-0.0 would come back as +0.0)
                    out[i] = j | (iv << 16);
                }
                if (!ok) { bad.store(1); return; }
            }
            done[k].fetch_add(1, std::memory_order_release);
        }
    };
    PackShare share{indptr, indices, data, n_cells, nnz, n_genes, f16, chunk, nchunks, T, pin, done.data(), &bad, &listed, getpid()};
    g_share = &share;
    share_lock.unlock();
    const double t_in = clk();
    if (shm_serve) {                      // the node's other ranks may follow from now on
        ShmHeader* h = shm->hdr;
        h->n_cells = n_cells; h->nnz = nnz; h->chunk = chunk; h->nchunks = nchunks; h->n_genes = n_genes; h->f16 = f16 ? 1 : 0; h->fingerprint = fp;
        for (int64_t k = 0; k < nchunks; ++k) h->esc_end[k].store(-1);
        h->bad.store(0);
        h->generation.store(gen, std::memory_order_release);
        shm_pub.done = true;
    }
    (void)wait_stream(ctx);               // the copies must not overtake whatever the main stream still does with the device buffers
    pool->start(worker);
    int rc = send_packed(ctx, share, true, t_in, shm_serve ? shm : nullptr);
    pool->wait();
    if (rc == DDX_OK && bad.load()) rc = 1;
    // close the job: nobody attaches any more; the pinned buffer, the lists and the counters stay until the attached contexts are done
    share_lock.lock();
    g_share = nullptr;
    g_share_cv.wait(share_lock, [&] { return share.readers == 0; });
    share_lock.unlock();
    if (rc == DDX_E_HIP) return set_err(ctx, DDX_E_HIP, "packed upload failed");
    if (rc != DDX_OK) { ctx->hvg_rows = -1; release(ctx, ctx->raw_packed); }      // not a packable matrix: the caller sends it plain, the packed bytes go back
    else ctx->upload_form = 1;
    return rc;
}

int ddx_pack_rows16(int64_t n_rows, const int64_t* indptr, const int32_t* indices, const float* data, uint16_t* codes, int64_t capacity,
                    int32_t* listed_pos, int32_t* listed_col, float* listed_val, int64_t* n_listed) {
    if (n_rows < 0 || !indptr || !n_listed) return set_err(nullptr, DDX_E_ARG, "bad arguments");
    DDX_TRY(check_csr(nullptr, n_rows, INT32_MAX, indptr, indices, data));
    const int64_t nnz = indptr[n_rows];
    if (nnz && !codes) return set_err(nullptr, DDX_E_ARG, "null output");
    std::vector<PackEsc> esc;
    if (nnz) {
        if (__builtin_cpu_supports("avx2")) pack16_avx2(indptr, n_rows, indices, data, 0, nnz, codes, esc);
        else pack16_generic(indptr, n_rows, indices, data, 0, nnz, codes, esc);
    }
    *n_listed = (int64_t)esc.size();
    const int64_t keep = std::min<int64_t>(capacity, (int64_t)esc.size());
    for (int64_t t = 0; t < keep; ++t) {
        if (listed_pos) listed_pos[t] = esc[t].pos;
        if (listed_col) listed_col[t] = esc[t].col;
        if (listed_val) listed_val[t] = esc[t].val;
    }
    return DDX_OK;
}

int ddx_set_upload_share(const char* name, int32_t local_rank, int32_t local_world) {
    return ddx_set_upload_share_impl(name, local_rank, local_world);
}

int ddx_set_upload_threads(int32_t n) {
    if (n < 0 || n > 1024) return set_err(nullptr, DDX_E_ARG, "thread count out of range");
    g_upload_threads.store(n);
    return DDX_OK;
}

int ddx_upload_raw(ddx_ctx* ctx, int64_t n_cells, int32_t n_genes, const int64_t* indptr, const int32_t* indices,
                   const float* data) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_TRY(check_csr(ctx, n_cells, n_genes, indptr, indices, data));
    int64_t nnz = indptr[n_cells];
    if (nnz >= (int64_t)1 << 31) return set_err(ctx, DDX_E_UNSUPPORTED, "more than 2^31-1 stored entries");
    context_reset(ctx);
    // everything a fit allocates on this context, in one chunk: raw CSR + HVG temporaries + restricted matrix, its
    // mirror, sort space, PCA / kNN / graph work space (about 95 bytes per stored raw entry at the benchmark shapes)
    arena_hint(ctx, (size_t)nnz * 100 + (size_t)n_cells * 6000 + ((size_t)1 << 30));
    DDX_TRY(ensure(ctx, ctx->raw_indptr, sizeof(int64_t) * (n_cells + 1)));
    DDX_TRY(ensure(ctx, ctx->raw_indices, sizeof(int32_t) * (size_t)(nnz + 1)));
    DDX_TRY(ensure(ctx, ctx->raw_data, sizeof(float) * (size_t)(nnz + 1)));
    DDX_HIP(ctx, hipMemcpyAsync(ctx->raw_indptr.p, indptr, sizeof(int64_t) * (n_cells + 1), hipMemcpyHostToDevice,
                                ctx->stream));
    ctx->upload_form = 0;
    if (nnz) {
        int packed = ctx->opt.upload_packed ? upload_packed(ctx, n_cells, nnz, n_genes, indptr, indices, data) : 1;
        if (packed < 0) return packed;
        if (packed != DDX_OK) {
            DDX_HIP(ctx, hipMemcpyAsync(ctx->raw_indices.p, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice,
                                        ctx->stream));
            DDX_HIP(ctx, hipMemcpyAsync(ctx->raw_data.p, data, sizeof(float) * nnz, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ctx->rawN = 0;
    DDX_TRY(validate_csr(ctx, ctx->raw_indptr.as<int64_t>(), ctx->raw_indices.as<int32_t>(), ctx->raw_data.as<float>(), n_cells, n_genes));
    ctx->rawN = n_cells;
    ctx->rawG = n_genes;
    ctx->raw_nnz = nnz;
    ctx->h_raw_indptr.assign(indptr, indptr + n_cells + 1);
    return DDX_OK;
}

int ddx_gene_variances(ddx_ctx* ctx, float* var_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    if (!ctx->rawN) return set_err(ctx, DDX_E_ARG, "ddx_upload_raw has not been called");
    if (!var_out) return set_err(ctx, DDX_E_ARG, "null output");
    return stage_gene_variances(ctx, var_out);
}

int ddx_select_columns(ddx_ctx* ctx, const int64_t* cols, int32_t n_cols) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    if (!ctx->rawN) return set_err(ctx, DDX_E_ARG, "ddx_upload_raw has not been called");
    if (!cols || n_cols <= 0) return set_err(ctx, DDX_E_ARG, "empty column selection");
    return stage_select_columns(ctx, cols, n_cols);
}

int ddx_upload_counts(ddx_ctx* ctx, int64_t n_cells, int32_t n_genes, const int64_t* indptr, const int32_t* indices,
                      const float* data) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_TRY(check_csr(ctx, n_cells, n_genes, indptr, indices, data));
    context_reset(ctx);
    return stage_upload_counts(ctx, n_cells, n_genes, indptr, indices, data, false);
}

#define NEED(cond, msg) \
    if (!(cond)) return set_err(ctx, DDX_E_ARG, msg)

int ddx_clone_counts(ddx_ctx* ctx, ddx_ctx* src) {
    REQUIRE_CTX(ctx);
    NEED(src && src != ctx, "source context must be another context");
    if (src->device != ctx->device)
        return set_err(ctx, DDX_E_UNSUPPORTED, "ddx_clone_counts: contexts live on different GPUs (%d, %d)", ctx->device, src->device);
    USE_DEVICE(ctx);
    // (no wait on the source's stream: what is copied was complete when the source published it -- CloneView --, and the source may be
    // running its own iterations on another host thread by now)
    context_reset(ctx);
    return stage_clone_counts(ctx, src);
}

static int d2h(ddx_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return DDX_OK;
    DDX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    return DDX_OK;
}

int ddx_get_counts_nnz(ddx_ctx* ctx, int64_t* nnz) {
    REQUIRE_CTX(ctx);
    NEED(ctx->have_counts, "no counts uploaded");
    *nnz = ctx->nnz;
    return DDX_OK;
}

int ddx_get_counts(ddx_ctx* ctx, int64_t* indptr, int32_t* indices, float* data) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_counts, "no counts uploaded");
    DDX_TRY(d2h(ctx, indptr, ctx->aug_indptr.p, sizeof(int64_t) * (ctx->N + 1)));
    DDX_TRY(d2h(ctx, indices, ctx->aug_indices.p, sizeof(int32_t) * ctx->nnz));
    DDX_TRY(d2h(ctx, data, ctx->aug_raw.p, sizeof(float) * ctx->nnz));
    return DDX_OK;
}

int ddx_get_lib_size(ddx_ctx* ctx, float* lib_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_counts, "no counts uploaded");
    return d2h(ctx, lib_out, ctx->lib32.p, sizeof(float) * ctx->N);
}

// normed values are recomputed on the fly inside the fused normalise kernel; this helper evaluates
// the same expression (float)((double)v / rowsum) on the host from device data for the parity test.
int ddx_get_normed(ddx_ctx* ctx, float* normed_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_counts, "no counts uploaded");
    std::vector<float> v(ctx->nnz);
    std::vector<double> l(ctx->N);
    DDX_TRY(d2h(ctx, v.data(), ctx->aug_raw.p, sizeof(float) * ctx->nnz));
    DDX_TRY(d2h(ctx, l.data(), ctx->lib64.p, sizeof(double) * ctx->N));
    for (int64_t i = 0; i < ctx->N; ++i) {
        double s = l[i];
        for (int64_t e = ctx->h_indptr[i]; e < ctx->h_indptr[i + 1]; ++e)
            normed_out[e] = (s == 0.0) ? v[e] : (float)((double)v[e] / s);
    }
    return DDX_OK;
}

// ---- doublets ---------------------------------------------------------------------------------
int ddx_create_doublets(ddx_ctx* ctx, int64_t n_synth, const int64_t* parents) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_counts, "no counts uploaded");
    NEED(n_synth >= 0, "negative n_synth");
    NEED(n_synth == 0 || parents, "null parents");
    for (int64_t i = 0; i < 2 * n_synth; ++i)
        if (parents[i] < 0 || parents[i] >= ctx->N)
            return set_err(ctx, DDX_E_ARG, "parent index %lld out of range at %lld", (long long)parents[i], (long long)i);
    return stage_create_doublets(ctx, n_synth, parents);
}

int ddx_get_synth_nnz(ddx_ctx* ctx, int64_t* nnz) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_synth, "no synthetic doublets");
    DDX_TRY(ensure_full_rows(ctx));
    int64_t last = 0;
    DDX_TRY(d2h(ctx, &last, ctx->aug_indptr.as<int64_t>() + ctx->M, sizeof(int64_t)));
    *nnz = last - ctx->nnz;
    return DDX_OK;
}

int ddx_get_synth(ddx_ctx* ctx, int64_t* indptr, int32_t* indices, float* data) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_synth, "no synthetic doublets");
    DDX_TRY(ensure_full_rows(ctx));
    DDX_TRY(d2h(ctx, indptr, ctx->aug_indptr.as<int64_t>() + ctx->N, sizeof(int64_t) * (ctx->S + 1)));
    int64_t base = ctx->nnz;
    int64_t n = indptr[ctx->S] - base;
    for (int64_t i = 0; i <= ctx->S; ++i) indptr[i] -= base;
    DDX_TRY(d2h(ctx, indices, ctx->aug_indices.as<int32_t>() + base, sizeof(int32_t) * n));
    DDX_TRY(d2h(ctx, data, ctx->aug_raw.as<float>() + base, sizeof(float) * n));
    return DDX_OK;
}

// ---- normalisation -----------------------------------------------------------------------------
int ddx_lognormalise(ddx_ctx* ctx, float pseudocount) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_synth, "ddx_create_doublets must run first");
    NEED(pseudocount > 0.f, "pseudocount must be positive");
    return stage_lognormalise(ctx, pseudocount);
}

int ddx_get_aug_lib(ddx_ctx* ctx, float* lib_out, float* median_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    if (lib_out) DDX_TRY(d2h(ctx, lib_out, ctx->lib32.p, sizeof(float) * ctx->M));
    if (median_out) DDX_TRY(d2h(ctx, median_out, ctx->median.p, sizeof(float)));
    return DDX_OK;
}

int ddx_get_aug_nnz(ddx_ctx* ctx, int64_t* nnz) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_synth, "no synthetic doublets");
    DDX_TRY(ensure_full_rows(ctx));
    return d2h(ctx, nnz, ctx->aug_indptr.as<int64_t>() + ctx->M, sizeof(int64_t));
}

int ddx_get_aug_values(ddx_ctx* ctx, float* values_out, float* zero_value_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    DDX_TRY(ensure_full_rows(ctx));
    int64_t n = 0;
    DDX_TRY(d2h(ctx, &n, ctx->aug_indptr.as<int64_t>() + ctx->M, sizeof(int64_t)));
    if (values_out) DDX_TRY(d2h(ctx, values_out, ctx->aug_x.p, sizeof(float) * n));
    if (zero_value_out) DDX_TRY(d2h(ctx, zero_value_out, ctx->zcol.p, sizeof(float) * ctx->H));
    return DDX_OK;
}

int ddx_get_aug_dense_rows(ddx_ctx* ctx, int64_t row0, int64_t nrows, float* out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    DDX_TRY(ensure_full_rows(ctx));
    NEED(row0 >= 0 && nrows >= 0 && row0 + nrows <= ctx->M, "row range out of bounds");
    if (!nrows) return DDX_OK;
    return stage_dense_rows(ctx, row0, nrows, out);
}

int ddx_scale(ddx_ctx* ctx, float max_value) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    NEED(!ctx->scaled, "matrix already scaled");
    return stage_scale(ctx, max_value);
}

// ---- PCA ----------------------------------------------------------------------------------------
int ddx_pca(ddx_ctx* ctx, int32_t n_components, int32_t n_oversamples, int32_t n_iter, const double* q0,
            int64_t q0_rows) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    NEED(n_components >= 1 && n_oversamples >= 0, "bad sketch size");
    int64_t want = (ctx->M >= ctx->H) ? (int64_t)ctx->H : ctx->M;
    if (!q0) {   // reuse the start matrix of the previous call (boosting iterations share their seeded start)
        NEED(ctx->q0_rows == want && ctx->q0_cols == n_components + n_oversamples, "no start matrix of this shape is resident");
    }
    if (q0_rows != want)
        return set_err(ctx, DDX_E_ARG, "q0 must have %lld rows (M=%lld, H=%d), got %lld", (long long)want,
                       (long long)ctx->M, ctx->H, (long long)q0_rows);
    int64_t mn = ctx->M < ctx->H ? ctx->M : (int64_t)ctx->H;
    if (n_components > mn) return set_err(ctx, DDX_E_ARG, "n_components=%d exceeds min(M,H)=%lld", n_components, (long long)mn);
    return stage_pca(ctx, n_components, n_oversamples, n_iter, q0, q0_rows);
}

int ddx_operator_apply(ddx_ctx* ctx, int32_t transpose, const double* X, int32_t n, double* out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise must run first");
    NEED(X && out, "null buffer");
    NEED(n >= 1 && n <= 64, "n must be in [1,64]");
    return stage_operator_apply(ctx, transpose, X, n, out);
}

int ddx_get_embedding(ddx_ctx* ctx, float* emb_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_emb, "no embedding");
    return d2h(ctx, emb_out, ctx->emb32.p, sizeof(float) * ctx->embM * ctx->C);
}

int ddx_get_embedding_f64(ddx_ctx* ctx, double* emb_out, double* singular_values) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_emb && ctx->emb64.p && ctx->sing.p, "no float64 embedding (set_embedding was used)");
    if (emb_out) DDX_TRY(d2h(ctx, emb_out, ctx->emb64.p, sizeof(double) * ctx->embM * ctx->C));
    if (singular_values) DDX_TRY(d2h(ctx, singular_values, ctx->sing.p, sizeof(double) * ctx->C));
    return DDX_OK;
}

int ddx_set_embedding(ddx_ctx* ctx, const float* emb, int64_t n_rows, int32_t n_components) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(emb && n_rows > 0 && n_components > 0, "bad embedding");
    DDX_TRY(ensure(ctx, ctx->emb32, sizeof(float) * n_rows * n_components));
    DDX_HIP(ctx, hipMemcpyAsync(ctx->emb32.p, emb, sizeof(float) * n_rows * n_components, hipMemcpyHostToDevice,
                                ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    ctx->embM = n_rows;
    ctx->C = n_components;
    ctx->have_emb = true;
    ctx->have_knn = false;
    return DDX_OK;
}

// ---- kNN / graph -------------------------------------------------------------------------------
int ddx_knn(ddx_ctx* ctx, int32_t k, int32_t include_self) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_emb, "no embedding");
    NEED(k >= 1 && k <= 256, "k must be in [1,256]");
    if ((int64_t)k + (include_self ? 0 : 1) > ctx->embM)
        return set_err(ctx, DDX_E_ARG, "k=%d too large for %lld points", k, (long long)ctx->embM);
    return stage_knn(ctx, k, include_self);
}

int ddx_knn_metric(ddx_ctx* ctx, int32_t k, int32_t include_self, int32_t metric) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_emb, "no embedding");
    NEED(k >= 1 && k <= 256, "k must be in [1,256]");
    NEED(metric >= 0 && metric <= 3, "metric must be 0 (euclidean), 1 (manhattan), 2 (cosine) or 3 (correlation)");
    if ((int64_t)k + (include_self ? 0 : 1) > ctx->embM)
        return set_err(ctx, DDX_E_ARG, "k=%d too large for %lld points", k, (long long)ctx->embM);
    if (metric == 0) return stage_knn(ctx, k, include_self);
    return stage_knn_metric(ctx, k, include_self, metric);
}

int ddx_get_knn(ddx_ctx* ctx, int32_t* idx_out, double* dist_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn, "no kNN result");
    if (idx_out) DDX_TRY(d2h(ctx, idx_out, ctx->knn_idx.p, sizeof(int32_t) * ctx->embM * ctx->K));
    if (dist_out) DDX_TRY(d2h(ctx, dist_out, ctx->knn_dist.p, sizeof(double) * ctx->embM * ctx->K));
    return DDX_OK;
}

int ddx_get_knn_window_fraction(ddx_ctx* ctx, double* fraction) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn && fraction, "no kNN result");
    if (!ctx->knn_window_total) { *fraction = 1.0; return DDX_OK; }      // (the scan of the other metrics meets every pair)
    unsigned long long total = 0;
    DDX_TRY(d2h(ctx, &total, ctx->knn_window_total, sizeof(total)));
    *fraction = ctx->knn_window_pairs > 0.0 ? (double)total / ctx->knn_window_pairs : 0.0;
    return DDX_OK;
}

int ddx_get_knn_overflow_count(ddx_ctx* ctx, int64_t* n_queries) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn && n_queries, "no kNN result");
    int32_t n = 0;
    if (ctx->knn_overflow) DDX_TRY(d2h(ctx, &n, ctx->knn_overflow, sizeof(n)));
    *n_queries = n;
    return DDX_OK;
}

int ddx_get_upload_form(ddx_ctx* ctx, int32_t* form) {
    REQUIRE_CTX(ctx);
    NEED(form, "null output");
    *form = ctx->upload_form;
    return DDX_OK;
}

int ddx_get_bitplane_stats(ddx_ctx* ctx, int64_t* out) {
    REQUIRE_CTX(ctx);
    NEED(out, "null output");
    const bool on = ctx->bp.ready && ctx->bp.values;
    out[0] = on ? 1 : 0;
    out[1] = on ? ctx->bp.nrest_o : 0;
    out[2] = on ? ctx->bp.nrest_s : 0;
    out[3] = ctx->opt.bp_digits == 3 ? 3 : 4;
    out[4] = (on && ctx->bp.scaled) ? 1 : 0;
    out[5] = ctx->bp.ready ? ctx->bp.n_demoted : 0;
    out[6] = ctx->opt.bp_mx ? 1 : 0;
    // digits of the power iterations before the last one (stage_pca's rule; 0: the same as out[3])
    const int early = ctx->opt.bp_digits_early;
    out[7] = (early && early < out[3] && !ctx->opt.bp_mx) ? early : 0;
    return DDX_OK;
}

int ddx_pca_exact_sparse(ddx_ctx* ctx, int32_t n_components, int32_t n_oversamples, double tol, int32_t max_steps, const double* start,
                         int32_t* steps_out, ddx_eigh_fn eigh, void* eigh_user) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_lognorm, "ddx_lognormalise first");
    NEED(start && n_components >= 1 && n_oversamples >= 0 && tol > 0.0, "ddx_pca_exact_sparse: start matrix, n_components >= 1, tol > 0");
    NEED((int64_t)n_components + n_oversamples <= std::min<int64_t>(ctx->M, ctx->H), "more vectors than the matrix has rows / columns");
    return stage_pca_block_lanczos(ctx, n_components, n_oversamples, tol, max_steps, start, steps_out, eigh, eigh_user);
}

int ddx_set_option(ddx_ctx* ctx, const char* key, const char* value) {
    REQUIRE_CTX(ctx);
    if (!key) return set_err(ctx, DDX_E_ARG, "ddx_set_option: no key");
    if (!ctx->opt.set(key, value)) return set_err(ctx, DDX_E_ARG, "ddx_set_option: unknown key or value '%s' = '%s'", key, value ? value : "");
    return DDX_OK;
}

int ddx_get_knn_candidate_counts(ddx_ctx* ctx, int32_t* counts_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn && counts_out, "no kNN result");
    return stage_knn_candidate_counts(ctx, counts_out);
}

int ddx_build_graph(ddx_ctx* ctx, int32_t mode) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn, "no kNN result");
    NEED(mode >= 0 && mode <= 3, "graph mode must be 0, 1, 2 or 3");
    if (mode >= 2) { NEED(ctx->knn_self, "modes 2 and 3 expect a kNN table computed with include_self=1"); }
    else { NEED(!ctx->knn_self, "Jaccard graphs expect a kNN table computed with include_self=0"); }
    return stage_build_graph(ctx, mode);
}

int ddx_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_out, double* w_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->have_knn, "no kNN result");
    NEED(mode >= 0 && mode <= 3, "graph mode must be 0, 1, 2 or 3");
    NEED(idx_out && w_out, "null output");
    if (mode >= 2) { NEED(ctx->knn_self, "modes 2 and 3 expect a kNN table computed with include_self=1"); }
    else { NEED(!ctx->knn_self, "Jaccard graphs expect a kNN table computed with include_self=0"); }
    return stage_graph_relations(ctx, mode, idx_out, w_out);
}

int ddx_assemble_graph(int64_t n_nodes, int32_t k, const int32_t* idx, const double* w, int64_t* indptr_out,
                       int32_t* indices_out, double* weights_out) {
    if (n_nodes < 0 || k <= 0 || !idx || !w || !indptr_out || !indices_out || !weights_out)
        return set_err(nullptr, DDX_E_ARG, "bad arguments to ddx_assemble_graph");
    for (int64_t t = 0; t < n_nodes * k; ++t)
        if (w[t] != 0.0 && (idx[t] < 0 || idx[t] >= n_nodes)) return set_err(nullptr, DDX_E_ARG, "neighbour index out of range");
    std::vector<int64_t> ip;
    std::vector<int32_t> gi;
    std::vector<double> gw;
    assemble_graph(n_nodes, k, idx, w, ip, gi, gw);
    memcpy(indptr_out, ip.data(), sizeof(int64_t) * ip.size());
    if (!gi.empty()) {
        memcpy(indices_out, gi.data(), sizeof(int32_t) * gi.size());
        memcpy(weights_out, gw.data(), sizeof(double) * gw.size());
    }
    return DDX_OK;
}

int ddx_get_graph_size(ddx_ctx* ctx, int64_t* n_nodes, int64_t* n_entries) {
    REQUIRE_CTX(ctx);
    NEED(ctx->g_nodes >= 0, "no graph");
    *n_nodes = ctx->g_nodes;
    *n_entries = ctx->g_entries;
    return DDX_OK;
}

int ddx_get_graph(ddx_ctx* ctx, int64_t* indptr, int32_t* indices, double* weights) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->g_nodes >= 0, "no graph");
    DDX_HIP(ctx, hipMemcpyAsync(indptr, ctx->g_d_indptr, sizeof(int64_t) * (ctx->g_nodes + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->g_entries > 0) {
        DDX_HIP(ctx, hipMemcpyAsync(indices, ctx->g_d_cols, sizeof(int32_t) * ctx->g_entries, hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, hipMemcpyAsync(weights, ctx->g_d_vals, sizeof(double) * ctx->g_entries, hipMemcpyDeviceToHost, ctx->stream));
    }
    DDX_HIP(ctx, wait_stream(ctx));
    return DDX_OK;
}

int ddx_coarsen_graph(ddx_ctx* ctx, double gamma, int32_t sweeps, int32_t levels) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->g_nodes >= 0, "no graph: call ddx_build_graph first");
    NEED(sweeps >= 0 && levels >= 1, "sweeps must be >= 0 and levels >= 1");
    return stage_coarsen_graph(ctx, gamma, sweeps, levels);
}

int ddx_refine_communities(ddx_ctx* ctx, const int32_t* coarse_labels, double gamma, int32_t sweeps, int32_t* labels_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->g_nodes >= 0 && ctx->c_nodes >= 0, "no graph / coarse graph on the device: ddx_build_graph and ddx_coarsen_graph come first");
    NEED(coarse_labels && labels_out && sweeps >= 0, "bad arguments");
    return stage_refine_communities(ctx, coarse_labels, gamma, sweeps, labels_out);
}

int ddx_get_coarse_size(ddx_ctx* ctx, int64_t* n_coarse, int64_t* n_entries) {
    REQUIRE_CTX(ctx);
    NEED(ctx->c_nodes >= 0, "no coarse graph");
    NEED(n_coarse && n_entries, "null output");
    *n_coarse = ctx->c_nodes;
    *n_entries = ctx->c_entries;
    return DDX_OK;
}

int ddx_get_coarse_graph(ddx_ctx* ctx, int32_t* member, int64_t* indptr, int32_t* indices, double* weights) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    NEED(ctx->c_nodes >= 0 && ctx->g_nodes >= 0 && ctx->lv_host_valid, "no coarse graph");
    DDX_HIP(ctx, wait_stream(ctx));        // the packed copy issued by ddx_coarsen_graph has landed
    const int64_t E = ctx->c_entries, nc = ctx->c_nodes, n = ctx->g_nodes;
    const double* hw = static_cast<const double*>(ctx->lv_host);
    const int64_t* hi = reinterpret_cast<const int64_t*>(hw + E);
    const int32_t* hm = reinterpret_cast<const int32_t*>(hi + nc + 1);
    const int32_t* hc = hm + n;
    std::memcpy(member, hm, sizeof(int32_t) * (size_t)n);
    std::memcpy(indptr, hi, sizeof(int64_t) * (size_t)(nc + 1));
    if (E > 0) {
        std::memcpy(indices, hc, sizeof(int32_t) * (size_t)E);
        std::memcpy(weights, hw, sizeof(double) * (size_t)E);
    }
    return DDX_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int ddx_timing_enable(ddx_ctx* ctx, int32_t on) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    if (!on) DDX_TRY(timing_flush(ctx));
    ctx->timing = on != 0;
    return DDX_OK;
}

int ddx_timing_reset(ddx_ctx* ctx) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_TRY(timing_flush(ctx));
    for (auto& r : ctx->t_recs) r = TimingRec();
    ctx->t_intervals.clear();
    return DDX_OK;
}

int ddx_timing_reference(ddx_ctx* ctx, ddx_ctx* share_with) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    if (share_with && share_with != ctx) {                 // the same clock origin as another context of this GPU
        NEED(share_with->device == ctx->device && share_with->t_ref, "the other context has no reference on this device");
        ctx->t_ref = share_with->t_ref;
        return DDX_OK;
    }
    hipEvent_t e = nullptr;                                // a new origin (contexts that shared the old one keep it)
    DDX_HIP(ctx, hipEventCreate(&e));
    ctx->t_ref = std::shared_ptr<ihipEvent_t>(e, [](hipEvent_t x) { (void)hipEventDestroy(x); });
    DDX_HIP(ctx, hipEventRecord(e, ctx->stream));
    DDX_HIP(ctx, hipEventSynchronize(e));
    return DDX_OK;
}

int ddx_timing_intervals(ddx_ctx* ctx, int64_t capacity, double* begin_end_ms, int64_t* n_out) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_TRY(timing_flush(ctx));
    const int64_t n = (int64_t)ctx->t_intervals.size() / 2;
    if (n_out) *n_out = n;
    if (begin_end_ms)
        for (int64_t i = 0; i < std::min(n, capacity) * 2; ++i) begin_end_ms[i] = (double)ctx->t_intervals[i];
    return DDX_OK;
}

int ddx_timing_count(ddx_ctx* ctx, int32_t* n) {
    REQUIRE_CTX(ctx);
    USE_DEVICE(ctx);
    DDX_TRY(timing_flush(ctx));
    *n = (int32_t)ctx->t_names.size();
    return DDX_OK;
}

int ddx_timing_get(ddx_ctx* ctx, int32_t i, char* name_out, int64_t* launches, double* total_ms) {
    REQUIRE_CTX(ctx);
    if (i < 0 || i >= (int32_t)ctx->t_names.size()) return set_err(ctx, DDX_E_ARG, "timing index out of range");
    snprintf(name_out, 64, "%s", ctx->t_names[i].c_str());
    *launches = ctx->t_recs[i].launches;
    *total_ms = ctx->t_recs[i].total_ms;
    return DDX_OK;
}

}  // extern "C"
