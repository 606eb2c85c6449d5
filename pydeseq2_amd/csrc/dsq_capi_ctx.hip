// dsq_capi_ctx.hip — context, memory, streams and the count upload of the C ABI (include/deseq_hip.h).
#include "dsq_capi_internal.h"

std::atomic<unsigned long long> g_dsq_host_syncs{0};

extern "C" {


// ------------------------------------------------------------------ context
int dsq_create(int device_id, dsq_ctx** out) {
    if (!out) return DSQ_ERR_ARG;
    *out = nullptr;
    dsq_ctx* ctx = new dsq_ctx();
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evk0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evk1);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_scratch, kScratchBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counter, 64);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_pin, 65536, hipHostMallocDefault);
    if (e != hipSuccess) {
        fprintf(stderr, "dsq_create: %s\n", hipGetErrorString(e));
        delete ctx;
        return DSQ_ERR_HIP;
    }
    *out = ctx;
    return DSQ_OK;
}

int dsq_set_deferred(dsq_ctx* ctx, int on) {
    ctx->deferred = on ? 1 : 0;
    return DSQ_OK;
}

void dsq_destroy(dsq_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->d_list) (void)hipFree(ctx->d_list);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_resume) (void)hipFree(ctx->d_resume);
    if (ctx->d_mix) (void)hipFree(ctx->d_mix);
    if (ctx->d_mixw) (void)hipFree(ctx->d_mixw);
    if (ctx->d_redo) (void)hipFree(ctx->d_redo);
    if (ctx->d_lfc_aux) (void)hipFree(ctx->d_lfc_aux);
    if (ctx->lfc_stream) (void)hipStreamDestroy(ctx->lfc_stream);
    if (ctx->ev_lfc_fork) (void)hipEventDestroy(ctx->ev_lfc_fork);
    if (ctx->ev_lfc_done) (void)hipEventDestroy(ctx->ev_lfc_done);
    if (ctx->ev_lfc_part) (void)hipEventDestroy(ctx->ev_lfc_part);
    dsq_internal_destroy_plugin(ctx);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_lsf) (void)hipFree(ctx->d_lsf);
    if (ctx->d_trend_grid) (void)hipFree(ctx->d_trend_grid);
    if (ctx->d_sum) (void)hipFree(ctx->d_sum);
    for (int k = 0; k < 2; ++k) {
        if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        if (ctx->d_stage16[k]) (void)hipFree(ctx->d_stage16[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->small_stream) (void)hipStreamDestroy(ctx->small_stream);
    if (ctx->ev_small0) (void)hipEventDestroy(ctx->ev_small0);
    if (ctx->ev_small1) (void)hipEventDestroy(ctx->ev_small1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evk0) (void)hipEventDestroy(ctx->evk0);
    if (ctx->evk1) (void)hipEventDestroy(ctx->evk1);
    if (ctx->main_stream) (void)hipStreamDestroy(ctx->main_stream);
    else if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* dsq_last_error(const dsq_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dsq_device_info(dsq_ctx* ctx, char* name, int name_len, int* cu_count, size_t* mem_bytes, char* arch,
                    int arch_len) {
    hipDeviceProp_t p;
    DSQ_HIP(hipGetDeviceProperties(&p, ctx->device));
    if (name && name_len > 0) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return DSQ_OK;
}

int dsq_sync(dsq_ctx* ctx) {
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_last_alpha_kernel(dsq_ctx* ctx, float* kernel_ms, int* n_grid_fallback) {
    if (kernel_ms) *kernel_ms = ctx->last_kernel_ms;
    if (n_grid_fallback) *n_grid_fallback = ctx->last_n_grid;
    return DSQ_OK;
}

// developer aid: name of the thread's pending (unconsumed) HIP error, "" if none; clears it
const char* dsq_debug_pending_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? "" : hipGetErrorString(e);
}

int dsq_timer_start(dsq_ctx* ctx) {
    DSQ_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return DSQ_OK;
}

int dsq_timer_stop(dsq_ctx* ctx, float* ms) {
    DSQ_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    DSQ_HIP(hipEventSynchronize(ctx->ev1));
    DSQ_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return DSQ_OK;
}

// ------------------------------------------------------------------ memory
int dsq_malloc(dsq_ctx* ctx, size_t bytes, void** dptr) {
    DSQ_CHECK_ARG(dptr != nullptr, "dsq_malloc: null out pointer");
    DSQ_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e == hipErrorOutOfMemory) return fail(ctx, DSQ_ERR_NOMEM, "hipMalloc: out of device memory");
    DSQ_HIP(e);
    return DSQ_OK;
}
int dsq_free(dsq_ctx* ctx, void* dptr) {
    if (dptr) DSQ_HIP(hipFree(dptr));
    return DSQ_OK;
}
int dsq_memset(dsq_ctx* ctx, void* dptr, int value, size_t bytes) {
    DSQ_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return DSQ_OK;
}
int dsq_h2d(dsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
    DSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_d2h(dsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
    DSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_h2d_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes,
               size_t rows) {
    DSQ_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_d2h_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes,
               size_t rows) {
    DSQ_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}


int dsq_side_begin(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->stream == ctx->main_stream || ctx->main_stream == nullptr, "already on the side stream");
    if (ctx->main_stream == nullptr) ctx->main_stream = ctx->stream;
    if (ctx->side_stream == nullptr) {
        // DSQ_CU_SPLIT=K: the side stream (robust dispersions: fills every CU it may use) is kept off K compute units
        // and the trend / prior kernels - 32 workgroups that synchronise through grid barriers - get a stream that may
        // use only those K: both then run at their stand-alone speed side by side, instead of the barrier kernel's
        // workgroups time-slicing CUs with the other kernel's waves (measured: 0.51 ms alone, 1.32 ms co-scheduled).
        // (c3: 9.10 -> 8.52 ms per step at K = 32, 8.58 at 64, no gain at 16 - two barrier workgroups per CU)
        // (read per context; the 32 was tuned on a 256-CU part: a device with fewer than 4 x split compute units keeps
        // one unmasked side stream - a robust-dispersion kernel squeezed onto cus - 32 units would serialise the stage)
        const int split = getenv("DSQ_CU_SPLIT") ? atoi(getenv("DSQ_CU_SPLIT")) : 32;
        hipDeviceProp_t prop;
        DSQ_HIP(hipGetDeviceProperties(&prop, ctx->device));
        const int cus = prop.multiProcessorCount;
        if (split > 0 && cus >= 4 * split) {
            std::vector<uint32_t> big((size_t)(cus + 31) / 32, 0u), small((size_t)(cus + 31) / 32, 0u);
            // which compute units are reserved: DSQ_CU_SPLIT_MODE 0 = the first `split` mask bits, 1 = every (cus / split)-th
            const int mode = getenv("DSQ_CU_SPLIT_MODE") ? atoi(getenv("DSQ_CU_SPLIT_MODE")) : 0;
            const int stride = cus / split;
            for (int i = 0; i < cus; ++i) {
                const bool res = mode == 0 ? i < split : (i % stride == 0 && i / stride < split);
                (res ? small : big)[(size_t)i / 32] |= 1u << (i % 32);
            }
            DSQ_HIP(hipExtStreamCreateWithCUMask(&ctx->side_stream, (uint32_t)big.size(), big.data()));
            DSQ_HIP(hipExtStreamCreateWithCUMask(&ctx->small_stream, (uint32_t)small.size(), small.data()));
            DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_small0, hipEventDisableTiming));
            DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_small1, hipEventDisableTiming));
        } else {
            DSQ_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        }
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    DSQ_HIP(hipEventRecord(ctx->ev_fork, ctx->main_stream));
    DSQ_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    ctx->stream = ctx->side_stream;
    return DSQ_OK;
}

int dsq_side_end(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->side_stream != nullptr && ctx->stream == ctx->side_stream, "not on the side stream");
    ctx->stream = ctx->main_stream;  // first: a failing record must not leave the context on the side stream
    DSQ_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
    return DSQ_OK;
}

// Leave the side stream whatever state the context is in (error paths of the caller: a stage failed between
// dsq_side_begin and dsq_side_end) and wait until everything queued on either stream has run, so that buffers the
// side stream was writing may be recycled.  A no-op on a context that never forked.
int dsq_side_abort(dsq_ctx* ctx) {
    if (ctx->lfc_stream != nullptr) {  // a forked LFC launch (dsq_lfc_fork_begin) whose partner will not come
        if (ctx->stream == ctx->lfc_stream && ctx->lfc_return != nullptr) ctx->stream = ctx->lfc_return;
        (void)hipStreamSynchronize(ctx->lfc_stream);
        ctx->lfc_pending_G = 0;
        ctx->lfc_part = nullptr; ctx->lfc_phase = 0;
    }
    if (ctx->main_stream != nullptr) ctx->stream = ctx->main_stream;
    if (ctx->side_stream != nullptr) {
        (void)hipEventRecord(ctx->ev_join, ctx->side_stream);
        DSQ_HIP(hipStreamSynchronize(ctx->side_stream));
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

// Fork for the first launch of an LFC fit in two (dsq_lfc_set_part): a stream of its own - every compute unit, unlike the
// side stream - that starts behind what the current stream holds now (the dispersion stage's full-size launch) and behind
// the side stream's last dsq_side_end (the robust dispersions its epilogue reads); calls made until dsq_lfc_fork_end go
// there.  The second launch of the fit (phase 2) joins it.
int dsq_lfc_fork_begin(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->lfc_stream == nullptr || ctx->stream != ctx->lfc_stream, "already forked");
    DSQ_CHECK_ARG(ctx->side_stream == nullptr || ctx->stream != ctx->side_stream, "on the side stream");
    DSQ_CHECK_ARG(ctx->lfc_pending_G == 0, "a forked LFC launch is still waiting for its second launch");
    if (ctx->lfc_stream == nullptr) {
        // lowest priority: the main stream's small kernels (the tail of the dispersion stage, one workgroup to a few hundred)
        // must find slots while this stream's full-size launch fills the device (at equal priority a 4 us kernel waited
        // 0.9 ms beside it)
        int least = 0, greatest = 0;
        DSQ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        static const bool flat = getenv("DSQ_LFC_FLAT_PRIORITY") != nullptr;  // A/B switch
        DSQ_HIP(hipStreamCreateWithPriority(&ctx->lfc_stream, hipStreamNonBlocking, flat ? (least + greatest) / 2 : least));
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_lfc_fork, hipEventDisableTiming));
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_lfc_done, hipEventDisableTiming));
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_lfc_part, hipEventDisableTiming));
    }
    DSQ_HIP(hipEventRecord(ctx->ev_lfc_fork, ctx->stream));
    DSQ_HIP(hipStreamWaitEvent(ctx->lfc_stream, ctx->ev_lfc_fork, 0));
    // (behind the side stream's work up to its last dsq_side_end: the caller forks before it puts more there)
    if (ctx->side_stream != nullptr) DSQ_HIP(hipStreamWaitEvent(ctx->lfc_stream, ctx->ev_join, 0));
    ctx->lfc_return = ctx->stream;
    ctx->stream = ctx->lfc_stream;
    return DSQ_OK;
}

int dsq_lfc_fork_end(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->lfc_stream != nullptr && ctx->stream == ctx->lfc_stream, "not forked");
    ctx->stream = ctx->lfc_return;  // first: a failing record must not leave the context on the forked stream
    DSQ_HIP(hipEventRecord(ctx->ev_lfc_done, ctx->lfc_stream));
    return DSQ_OK;
}

int dsq_side_wait(dsq_ctx* ctx) {
    if (ctx->side_stream == nullptr) return DSQ_OK;
    DSQ_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->ev_join, 0));
    return DSQ_OK;
}


int dsq_d2d(dsq_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return DSQ_OK;
}
// Host count matrix (int64 as the reference holds it, or int32) -> int32 in HBM, same element order.
// The matrix is cut into chunks of kStageElems elements; a few host threads narrow a chunk into one of two
// page-locked staging buffers (checking 0 <= v < 2^31) while the DMA of the previous chunk is in flight, so
// the PCIe link carries half the bytes of the int64 matrix and never waits for pageable-memory staging.
extern "C++" {
namespace {
constexpr size_t kStageElems = (size_t)8 << 20;  // 32 MiB of int32 per chunk

template <class SrcT>
void narrow_chunk(const SrcT* src, int32_t* dst, size_t n, int n_threads, int* bad) {
    auto work = [=](size_t lo, size_t hi, int* flag) {
        int b = 0;
        for (size_t i = lo; i < hi; ++i) {
            const SrcT v = src[i];
            b |= (v < 0) | ((long long)v > 2147483647LL);
            dst[i] = (int32_t)v;
        }
        if (b) *flag = 1;
    };
    if (n_threads <= 1 || n < ((size_t)1 << 16)) {
        work(0, n, bad);
        return;
    }
    std::vector<std::thread> th;
    std::vector<int> flags((size_t)n_threads, 0);
    const size_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        th.emplace_back(work, lo, hi, &flags[(size_t)t]);
    }
    for (auto& x : th) x.join();
    for (int f : flags)
        if (f) *bad = 1;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
// the same, to uint16 - for a chunk whose counts are all below 65 536 (the usual RNA-seq matrix): the PCIe link, which is
// what bounds the upload (240 MB of int32 at ~25 GB/s of pinned-memory DMA: 10 of the 12 ms), carries a QUARTER of the
// int64 matrix's bytes; the device widens the chunk into its place.  *big is set if some count does not fit (the chunk
// is then narrowed to int32 as before); negative counts set *bad.
template <class SrcT>
void narrow_chunk_u16(const SrcT* src, uint16_t* dst, size_t n, int n_threads, int* bad, int* big) {
    auto work = [=](size_t lo, size_t hi, int* flags) {
        int b = 0, g = 0;
        for (size_t i = lo; i < hi; ++i) {
            const SrcT v = src[i];
            b |= (v < 0);
            g |= ((long long)v > 65535LL);
            dst[i] = (uint16_t)v;
        }
        if (b) flags[0] = 1;
        if (g) flags[1] = 1;
    };
    std::vector<int> flags((size_t)2 * (n_threads > 1 ? n_threads : 1), 0);
    if (n_threads <= 1 || n < ((size_t)1 << 16)) {
        work(0, n, flags.data());
    } else {
        std::vector<std::thread> th;
        const size_t per = (n + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; ++t) {
            const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
            if (lo >= hi) break;
            th.emplace_back(work, lo, hi, &flags[(size_t)2 * t]);
        }
        for (auto& x : th) x.join();
    }
    for (size_t t = 0; t < flags.size(); t += 2) {
        if (flags[t]) *bad = 1;
        if (flags[t + 1]) *big = 1;
    }
}
}  // namespace
}  // extern "C++"

int dsq_upload_counts_i32(dsq_ctx* ctx, const void* counts, int count_type, size_t n_elems, int32_t* d_dst,
                          int* h_bad) {
    DSQ_CHECK_ARG(count_type == DSQ_I32 || count_type == DSQ_I64, "count_type");
    if (h_bad) *h_bad = 0;
    if (n_elems == 0) return DSQ_OK;
    for (int k = 0; k < 2; ++k) {
        if (!ctx->stage[k]) DSQ_HIP(hipHostMalloc(&ctx->stage[k], kStageElems * sizeof(int32_t), hipHostMallocDefault));
        if (!ctx->stage_ev[k]) DSQ_HIP(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    static const int n_threads = [] {
        const char* e = getenv("DSQ_UPLOAD_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency() / 2;
        const int cap = e ? 128 : 16;
        return t < 1 ? 1 : (t > cap ? cap : t);
    }();
    static const bool no_u16 = getenv("DSQ_UPLOAD_NO_U16") != nullptr;  // A/B switch
    int bad = 0;
    size_t off = 0;
    for (int c = 0; off < n_elems; ++c, off += kStageElems) {
        const size_t n = n_elems - off < kStageElems ? n_elems - off : kStageElems;
        const int k = c & 1;
        if (c >= 2) DSQ_HIP(hipEventSynchronize(ctx->stage_ev[k]));  // the DMA out of this buffer has finished
        int32_t* st = (int32_t*)ctx->stage[k];
        int big = no_u16 ? 1 : 0;
        if (!big) {  // optimistic: the chunk as uint16
            if (count_type == DSQ_I64) narrow_chunk_u16((const int64_t*)counts + off, (uint16_t*)st, n, n_threads, &bad, &big);
            else narrow_chunk_u16((const int32_t*)counts + off, (uint16_t*)st, n, n_threads, &bad, &big);
        }
        if (!big) {
            if (!ctx->d_stage16[k]) DSQ_HIP(hipMalloc(&ctx->d_stage16[k], kStageElems * sizeof(uint16_t)));
            // (stream order: the widening kernel of chunk c - 2 has read this device buffer before this copy starts)
            DSQ_HIP(hipMemcpyAsync(ctx->d_stage16[k], st, n * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
            DSQ_HIP(hipEventRecord(ctx->stage_ev[k], ctx->stream));
            DSQ_HIP(dsq::launch_widen_u16(ctx->stream, (const uint16_t*)ctx->d_stage16[k], d_dst + off, n));
        } else {
            if (count_type == DSQ_I64) narrow_chunk((const int64_t*)counts + off, st, n, n_threads, &bad);
            else narrow_chunk((const int32_t*)counts + off, st, n, n_threads, &bad);
            DSQ_HIP(hipMemcpyAsync(d_dst + off, st, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            DSQ_HIP(hipEventRecord(ctx->stage_ev[k], ctx->stream));
        }
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (h_bad) *h_bad = bad;
    return DSQ_OK;
}

// Page-locked host memory.  hipHostMalloc of 480 MB (one N x G output layer of the plug-in path at c3) takes 83-107 ms on the
// GPU box; ordinary memory whose pages a few threads touch first (4-5 ms) and hipHostRegister (1 ms) is page-locked in 6 ms
// and takes DMA at the same 56 GB/s from the second copy on (the first one 17 instead of 8.4 ms:
// tools/probes/register_probe2.py).  Large requests go that way; small ones, and a registration that fails (a low
// RLIMIT_MEMLOCK), take hipHostMalloc.  DSQ_HOST_ALLOC_MALLOC=1: hipHostMalloc always (A/B switch).
// (which buffers were registered rather than allocated by the runtime: process-wide, because a buffer may outlive the context
// it was made through - the Python pools give theirs back from finalisers)
static std::mutex g_host_mu;
static std::vector<void*> g_host_registered;

int dsq_host_alloc(dsq_ctx* ctx, size_t bytes, void** out) {
    DSQ_CHECK_ARG(out != nullptr, "null output pointer");
    DSQ_HIP(hipSetDevice(ctx->device));  // (callable from a helper thread)
    static const bool always_malloc = getenv("DSQ_HOST_ALLOC_MALLOC") != nullptr;
    constexpr size_t kMinRegister = (size_t)8 << 20, kPage = 4096, kAlign = (size_t)2 << 20;
    if (!always_malloc && bytes >= kMinRegister) {
        void* p = nullptr;
        const size_t padded = (bytes + kAlign - 1) & ~(kAlign - 1);
        if (posix_memalign(&p, kAlign, padded) == 0 && p != nullptr) {
            const int n_threads = 16;
            std::vector<std::thread> th;
            const size_t per = ((padded / kPage + n_threads - 1) / n_threads) * kPage;
            for (int t = 0; t < n_threads; ++t) {
                const size_t lo = (size_t)t * per, hi = lo + per < padded ? lo + per : padded;
                if (lo >= hi) break;
                th.emplace_back([=] { for (size_t o = lo; o < hi; o += kPage) ((volatile char*)p)[o] = 0; });
            }
            for (auto& x : th) x.join();
            const hipError_t e = hipHostRegister(p, padded, hipHostRegisterDefault);
            if (e == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_host_mu);
                g_host_registered.push_back(p);
                *out = p;
                return DSQ_OK;
            }
            (void)hipGetLastError();
            free(p);
        }
    }
    DSQ_HIP(hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
    return DSQ_OK;
}
int dsq_host_free(dsq_ctx* ctx, void* p) {
    if (p == nullptr) return DSQ_OK;
    bool registered = false;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (size_t i = 0; i < g_host_registered.size(); ++i)
            if (g_host_registered[i] == p) {
                g_host_registered[i] = g_host_registered.back();
                g_host_registered.pop_back();
                registered = true;
                break;
            }
    }
    if (registered) {
        (void)hipHostUnregister(p);  // (the context may be gone: no error report through it)
        (void)hipGetLastError();
        free(p);
        return DSQ_OK;
    }
    DSQ_HIP(hipHostFree(p));
    return DSQ_OK;
}
// asynchronous device -> pinned-host copy on the context's stream (pair with dsq_sync)
int dsq_d2h_async(dsq_ctx* ctx, void* pinned_dst, const void* d_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return DSQ_OK;
}
int dsq_h2d_async(dsq_ctx* ctx, void* d_dst, const void* pinned_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(d_dst, pinned_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DSQ_OK;
}


unsigned long long dsq_host_sync_count(void) { return g_dsq_host_syncs.load(); }

}  // extern "C"
