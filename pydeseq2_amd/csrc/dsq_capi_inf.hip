// dsq_capi_inf.hip — the Inference-level entry points (dsq_inf_*: the drop-in boundary) and their device cache.
#include <thread>

#include "dsq_capi_internal.h"

#include "dsq_plugin_cache.h"

// ================================================================== Inference-level API (the drop-in boundary)
// Host arrays in, host arrays out - one entry point per method of pydeseq2.inference.Inference.  The N x G matrices go
// through the content-addressed device cache of dsq_plugin_cache.h: the 7-9 calls of one deseq2() upload the counts, the
// normalised counts and nothing else; mu_hat / mu stay resident between the call that produces them and the calls that
// take them back.  Every device buffer of these calls comes from the cache's free list.
struct PluginDesign {
    struct One {
        std::vector<double> X;  // the host design this entry was built from (N x P row-major): compared byte for byte
        int N = 0, P = 0, ldx = 0, full_rank = 1;
        double *Xt = nullptr, *pinv = nullptr;  // device [P][ldx]
        dsq_mix* mix = nullptr;                 // mixed-design descriptor (NULL: not such a design); built on first request
        int mix_ready = 0;
        uint64_t tick = 0;
    };
    std::vector<One> v;
    uint64_t tick = 0;
};

void dsq_internal_destroy_plugin(dsq_ctx* ctx) {
    if (ctx->pc) {
        dsq_pc::destroy(*ctx->pc);
        delete ctx->pc;
        ctx->pc = nullptr;
    }
    if (ctx->designs == nullptr) return;
    for (auto& o : ctx->designs->v) {
        if (o.Xt) (void)hipFree(o.Xt);
        dsq_mix_destroy(o.mix);
    }
    delete ctx->designs;
    ctx->designs = nullptr;
}

namespace {

constexpr int kPluginDesigns = 4;

dsq_pc::Cache& plugin_cache(dsq_ctx* ctx) {
    if (ctx->pc == nullptr) {
        ctx->pc = new dsq_pc::Cache();
        dsq_pc::Cache& c = *ctx->pc;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
        c.budget = total_b / 4;  // (MI355X: 72 GB; the matrices of BASELINE configs[4] at full size are 8.4 GB)
        if (const char* e = getenv("DSQ_PLUGIN_CACHE_MB")) c.budget = (size_t)atoll(e) << 20;
        if (const char* e = getenv("DSQ_PLUGIN_CACHE")) c.enabled = atoi(e) != 0;
        int hw = (int)std::thread::hardware_concurrency();
        c.hash_threads = hw >= 64 ? 32 : (hw >= 4 ? hw / 2 : 1);
        if (const char* e = getenv("DSQ_HASH_THREADS")) c.hash_threads = std::max(1, atoi(e));
        c.verify = getenv("DSQ_PLUGIN_CACHE_VERIFY") != nullptr;
    }
    return *ctx->pc;
}

// a device buffer of the running call, back on the free list when the call returns (or adopted by the cache)
struct PcBuf {
    dsq_ctx* ctx = nullptr;
    void* p = nullptr;
    size_t cap = 0;
    PcBuf() = default;
    PcBuf(const PcBuf&) = delete;
    PcBuf& operator=(const PcBuf&) = delete;
    ~PcBuf() {
        if (p) dsq_pc::give(*ctx->pc, p, cap);
    }
    hipError_t alloc(dsq_ctx* c, size_t bytes) {
        ctx = c;
        return dsq_pc::take(plugin_cache(c), bytes, &p, &cap);
    }
    void release() { p = nullptr; }  // (ownership went to a cache entry)
    template <class T>
    T* as() { return (T*)p; }
};

int pc_begin(dsq_ctx* ctx) {
    DSQ_HIP(hipSetDevice(ctx->device));
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (c.d_acc == nullptr) {
        DSQ_HIP(hipMalloc((void**)&c.d_acc, 4 * sizeof(unsigned long long)));
        DSQ_HIP(hipHostMalloc((void**)&c.h_acc, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    }
    dsq_pc::begin_call(c);
    return DSQ_OK;
}

int pc_upload_small(dsq_ctx* ctx, const void* src, size_t bytes, PcBuf& dst) {
    DSQ_HIP(dst.alloc(ctx, bytes));
    if (bytes) DSQ_HIP(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DSQ_OK;
}

// verify mode: two gene-major device matrices ([G][ld_words] 32-bit words, n_words used per row) must agree word for word
int pc_verify(dsq_ctx* ctx, const void* fresh, const void* resident, int ld_words, int n_words, int G) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(dsq_pc::k_count_diff, dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream, (const uint32_t*)fresh,
                       (const uint32_t*)resident, ld_words, n_words, G, c.d_acc);
    DSQ_HIP(hipGetLastError());
    DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    ++c.st.verified;
    if (c.h_acc[3] != 0)
        return fail(ctx, DSQ_ERR_ARG, "plug-in cache: a matrix with the digest of a resident one differs from it "
                                           "(DSQ_PLUGIN_CACHE_VERIFY)");
    return DSQ_OK;
}

// host count matrix -> resident gene-major int32 [G][ldn] (the cache's, not to be freed by the caller)
int pc_counts(dsq_ctx* ctx, const void* counts, int count_type, int layout, int N, int G, int ldn,
              const int32_t** d_y, dsq_pc::Entry** ent = nullptr) {
    DSQ_CHECK_ARG(count_type == DSQ_I32 || count_type == DSQ_I64, "count_type");
    DSQ_CHECK_ARG(layout == DSQ_SAMPLE_MAJOR || layout == DSQ_GENE_MAJOR, "layout");
    dsq_pc::Cache& c = plugin_cache(ctx);
    dsq_pc::Timer t;
    const dsq_pc::Digest dg = count_type == DSQ_I64
                                  ? dsq_pc::digest_host((const int64_t*)counts, layout, N, G, c.hash_threads)
                                  : dsq_pc::digest_host((const int32_t*)counts, layout, N, G, c.hash_threads);
    c.st.hash_ms += t.ms();
    dsq_pc::Entry* hit = dsq_pc::find(c, dsq_pc::kCounts, N, G, dg);
    if (hit != nullptr && !c.verify) {
        ++c.st.hits;
        *d_y = (const int32_t*)hit->d;
        if (ent) *ent = hit;
        return DSQ_OK;
    }
    if (hit == nullptr) ++c.st.misses;
    PcBuf raw, y;
    DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(int32_t)));
    DSQ_HIP(y.alloc(ctx, (size_t)G * ldn * sizeof(int32_t)));
    int bad = 0, rc;
    if ((rc = dsq_upload_counts_i32(ctx, counts, count_type, (size_t)N * G, raw.as<int32_t>(), &bad))) return rc;
    if (bad) return fail(ctx, DSQ_ERR_RANGE, "counts must be integers in [0, 2^31)");
    c.st.h2d_bytes += (size_t)N * G * (count_type == DSQ_I64 ? 8 : 4);
    DSQ_HIP(dsq::launch_transpose_counts(ctx->stream, raw.p, DSQ_I32, layout, N, G, y.as<int32_t>(), ldn,
                                         (int*)ctx->d_scratch));
    if (hit != nullptr) {  // DSQ_PLUGIN_CACHE_VERIFY: the digest matched - do the bytes?
        int rc2;
        if ((rc2 = pc_verify(ctx, y.p, hit->d, ldn, N, G))) return rc2;
        ++c.st.hits;
        *d_y = (const int32_t*)hit->d;
        if (ent) *ent = hit;
        return DSQ_OK;
    }
    dsq_pc::Entry e;
    e.kind = dsq_pc::kCounts; e.N = N; e.G = G; e.ld = ldn; e.dg = dg; e.d = y.p; e.cap = y.cap;
    y.release();
    dsq_pc::Entry* ne = dsq_pc::insert(c, e);
    *d_y = (const int32_t*)ne->d;
    if (ent) *ent = ne;
    return DSQ_OK;
}

// host fp64 matrix -> resident gene-major [G][ldn]; need_positive: refuse a matrix with an element that is not positive,
// finite and normal (the reference's loss is inf / NaN there: y * log(mu), utils.py:227-234; the kernels' table logarithm
// is undefined) - checked once per resident matrix, on the device
int pc_f64(dsq_ctx* ctx, const double* src, int layout, int N, int G, int ldn, bool need_positive, const double** d_out) {
    DSQ_CHECK_ARG(layout == DSQ_SAMPLE_MAJOR || layout == DSQ_GENE_MAJOR, "layout");
    dsq_pc::Cache& c = plugin_cache(ctx);
    dsq_pc::Timer t;
    const dsq_pc::Digest dg = dsq_pc::digest_host(src, layout, N, G, c.hash_threads);
    c.st.hash_ms += t.ms();
    dsq_pc::Entry* e = dsq_pc::find(c, dsq_pc::kF64, N, G, dg);
    if (e != nullptr && c.verify) {  // DSQ_PLUGIN_CACHE_VERIFY: upload again and compare with the resident copy
        PcBuf raw, m;
        DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(double)));
        DSQ_HIP(m.alloc(ctx, (size_t)G * ldn * sizeof(double)));
        DSQ_HIP(hipMemcpyAsync(raw.p, src, (size_t)N * G * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(dsq::launch_transpose_f64(ctx->stream, raw.as<double>(), layout, N, G, m.as<double>(), ldn));
        int rc2;
        if ((rc2 = pc_verify(ctx, m.p, e->d, 2 * ldn, 2 * N, G))) return rc2;
    }
    if (e != nullptr) {
        ++c.st.hits;
    } else {
        ++c.st.misses;
        PcBuf raw, m;
        DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(double)));
        DSQ_HIP(m.alloc(ctx, (size_t)G * ldn * sizeof(double)));
        DSQ_HIP(hipMemcpyAsync(raw.p, src, (size_t)N * G * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        c.st.h2d_bytes += (size_t)N * G * sizeof(double);
        DSQ_HIP(dsq::launch_transpose_f64(ctx->stream, raw.as<double>(), layout, N, G, m.as<double>(), ldn));
        dsq_pc::Entry ne;
        ne.kind = dsq_pc::kF64; ne.N = N; ne.G = G; ne.ld = ldn; ne.dg = dg; ne.d = m.p; ne.cap = m.cap;
        m.release();
        e = dsq_pc::insert(c, ne);
    }
    if (need_positive && e->positive < 0) {
        DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL((dsq_pc::k_digest<double, true>), dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream,
                           (const double*)e->d, ldn, N, G, c.d_acc);
        DSQ_HIP(hipGetLastError());
        DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
        e->positive = c.h_acc[2] ? 0 : 1;
    }
    if (need_positive && e->positive == 0)
        return fail(ctx, DSQ_ERR_ARG,
                    "mu must be positive, finite and normal (the negative binomial log-likelihood takes log(mu))");
    *d_out = (const double*)e->d;
    return DSQ_OK;
}

// A gene-major fp64 matrix this call PRODUCED and has copied to the host stays resident under the digest of that host copy
// (computed here, on the device): the caller's next call that hands it back finds it.  Synchronises the stream.
int pc_adopt_f64(dsq_ctx* ctx, PcBuf& buf, int N, int G, int ldn) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (!c.enabled) return DSQ_OK;  // (the buffer goes back to the free list with its owner)
    DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL((dsq_pc::k_digest<double, true>), dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream,
                       buf.as<double>(), ldn, N, G, c.d_acc);
    DSQ_HIP(hipGetLastError());
    DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    dsq_pc::Digest dg;
    dg.a = c.h_acc[0]; dg.b = c.h_acc[1];
    if (dsq_pc::find(c, dsq_pc::kF64, N, G, dg) != nullptr) return DSQ_OK;  // (the same matrix is resident already)
    dsq_pc::Entry e;
    e.kind = dsq_pc::kF64; e.N = N; e.G = G; e.ld = ldn; e.dg = dg; e.d = buf.p; e.cap = buf.cap;
    e.positive = c.h_acc[2] ? 0 : 1;
    buf.release();
    dsq_pc::insert(c, e);
    ++c.st.adopted;
    return DSQ_OK;
}

// G rows of N doubles (device pitch ldn) -> a contiguous G x N host matrix.
// A page-locked destination (the layers of a caller's later fits, inference.py:_layer) takes one 2-D DMA.  A PAGEABLE one -
// the layers of the first fits: fresh memory, every page touched for the first time - took 37 ms per 480 MB that way
// (the runtime stages it through one thread: tools/probes/pin_probe.py).  Here: 32 MiB blocks of rows go by DMA into the two
// page-locked staging buffers of the upload path, and while the next block is in flight a few host threads copy the
// previous one to its place - the page faults of first touch spread over the threads.  DSQ_PLUGIN_D2H_THREADS=0: the
// runtime's own path (A/B switch).
int pc_download_rows(dsq_ctx* ctx, double* dst, const double* d_src, int ldn, int N, int G) {
    const size_t row = (size_t)N * sizeof(double), total = row * (size_t)G;
    plugin_cache(ctx).st.d2h_bytes += total;
    static const int n_threads = [] {
        const char* e = getenv("DSQ_PLUGIN_D2H_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency() / 2;
        const int cap = e ? 128 : 16;
        return t < 0 ? 0 : (t > cap ? cap : t);
    }();
    constexpr size_t kBlock = (size_t)32 << 20;
    bool pageable = false;
    if (n_threads > 0 && total >= 4 * kBlock && row <= kBlock) {
        hipPointerAttribute_t attr;
        const hipError_t e = hipPointerGetAttributes(&attr, dst);
        (void)hipGetLastError();  // (an unregistered host pointer is reported as an error: that is the answer)
        pageable = e != hipSuccess || attr.type == hipMemoryTypeUnregistered;
    }
    if (!pageable) {
        DSQ_HIP(hipMemcpy2DAsync(dst, row, d_src, (size_t)ldn * sizeof(double), row, (size_t)G, hipMemcpyDeviceToHost,
                                 ctx->stream));
        return DSQ_OK;
    }
    for (int k = 0; k < 2; ++k) {
        if (!ctx->stage[k]) DSQ_HIP(hipHostMalloc(&ctx->stage[k], kBlock, hipHostMallocDefault));
        if (!ctx->stage_ev[k]) DSQ_HIP(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    const size_t rows_per = kBlock / row;
    auto copy_out = [&](int k, size_t g0, size_t rows) {  // staging buffer k -> rows g0 .. g0 + rows of dst
        const char* src = (const char*)ctx->stage[k];
        char* d = (char*)dst + g0 * row;
        const size_t bytes = rows * row;
        const size_t per = ((bytes + (size_t)n_threads - 1) / (size_t)n_threads + 4095) & ~(size_t)4095;  // whole pages
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) {
            const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
            if (lo >= hi) break;
            th.emplace_back([=] { std::memcpy(d + lo, src + lo, hi - lo); });
        }
        for (auto& x : th) x.join();
    };
    size_t g0 = 0, prev_g0 = 0, prev_rows = 0;
    for (int c = 0; g0 < (size_t)G; ++c) {
        const size_t rows = (size_t)G - g0 < rows_per ? (size_t)G - g0 : rows_per;
        const int k = c & 1;
        // (buffer k was copied out two blocks ago, before this DMA is enqueued: the copies below are synchronous)
        DSQ_HIP(hipMemcpy2DAsync(ctx->stage[k], row, d_src + g0 * (size_t)ldn, (size_t)ldn * sizeof(double), row, rows,
                                 hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipEventRecord(ctx->stage_ev[k], ctx->stream));
        if (prev_rows != 0) {  // the previous block has landed (or lands now): copy it out while this one is in flight
            DSQ_HIP(hipEventSynchronize(ctx->stage_ev[k ^ 1]));
            copy_out(k ^ 1, prev_g0, prev_rows);
        }
        prev_g0 = g0; prev_rows = rows;
        g0 += rows;
        if (g0 >= (size_t)G) {
            DSQ_HIP(hipEventSynchronize(ctx->stage_ev[k]));
            copy_out(k, prev_g0, prev_rows);
        }
    }
    return DSQ_OK;
}

// the design of a call: factorised once per distinct matrix (QR start values, rank), its device copies and - on request -
// its mixed-design descriptor kept for the next calls (one deseq2() passes the same design 8 times)
int pc_design(dsq_ctx* ctx, const double* design, int N, int P, bool want_mix, PluginDesign::One** out) {
    if (ctx->designs == nullptr) ctx->designs = new PluginDesign();
    PluginDesign& D = *ctx->designs;
    PluginDesign::One* hit = nullptr;
    for (auto& o : D.v)
        if (o.N == N && o.P == P && std::memcmp(o.X.data(), design, (size_t)N * P * sizeof(double)) == 0) hit = &o;
    if (hit == nullptr) {
        if ((int)D.v.size() >= kPluginDesigns) {  // replace the least recently used one
            int lru = 0;
            for (int i = 1; i < (int)D.v.size(); ++i)
                if (D.v[(size_t)i].tick < D.v[(size_t)lru].tick) lru = i;
            DSQ_HIP(hipStreamSynchronize(ctx->stream));
            if (D.v[(size_t)lru].Xt) (void)hipFree(D.v[(size_t)lru].Xt);
            dsq_mix_destroy(D.v[(size_t)lru].mix);
            D.v.erase(D.v.begin() + lru);
        }
        PluginDesign::One o;
        o.N = N; o.P = P; o.ldx = pad16(N);
        o.X.assign(design, design + (size_t)N * P);
        std::vector<double> Xt, pinv;
        design_factor(design, N, P, o.ldx, Xt, pinv, o.full_rank);
        const size_t bytes = Xt.size() * sizeof(double);
        DSQ_HIP(hipMalloc((void**)&o.Xt, 2 * bytes));
        o.pinv = o.Xt + Xt.size();
        DSQ_HIP(hipMemcpyAsync(o.Xt, Xt.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(hipMemcpyAsync(o.pinv, pinv.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
        D.v.push_back(std::move(o));
        hit = &D.v.back();
    }
    hit->tick = ++D.tick;
    if (want_mix && !hit->mix_ready) {
        int rc;
        if ((rc = dsq_mix_create(ctx, design, N, P, &hit->mix))) return rc;
        hit->mix_ready = 1;
    }
    *out = hit;
    return DSQ_OK;
}

// gene lists of the mixed-design dispersion kernel for a resident count matrix (kept with the cache entry)
int pc_mix_lists(dsq_ctx* ctx, dsq_pc::Entry* e, int N, int G, int ldn) {
    if (e->lists_ready) return DSQ_OK;
    dsq_pc::Cache& c = plugin_cache(ctx);
    PcBuf flags;
    DSQ_HIP(flags.alloc(ctx, (size_t)G * sizeof(int32_t)));
    DSQ_HIP(dsq::launch_count_big(ctx->stream, (const int32_t*)e->d, ldn, N, G, flags.as<int32_t>()));
    std::vector<int32_t> fl((size_t)G), lists((size_t)G);
    DSQ_HIP(hipMemcpyAsync(fl.data(), flags.p, (size_t)G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    int nr = 0, nw = 0;  // rows first, waves from the end
    for (int g = 0; g < G; ++g) {
        if (fl[(size_t)g] >= 0) lists[(size_t)nr++] = g;
        else lists[(size_t)(G - 1 - nw++)] = g;
    }
    std::reverse(lists.begin() + nr, lists.end());
    void* d = nullptr;
    size_t cap = 0;
    DSQ_HIP(dsq_pc::take(c, (size_t)G * sizeof(int32_t), &d, &cap));
    DSQ_HIP(hipMemcpyAsync(d, lists.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));  // the host list goes out of scope
    e->d_lists = d; e->lists_cap = cap; e->n_rows = nr; e->n_waves = nw; e->lists_ready = 1;
    c.resident += cap;
    return DSQ_OK;
}

}  // namespace

extern "C" {

int dsq_abi_version(void) { return DSQ_ABI_VERSION; }

// the digest the cache identifies a host matrix by (no context, no GPU: CPU tests pin its layout / dtype independence)
int dsq_plugin_digest_host(const void* data, int elem_type, int layout, int N, int G, int n_threads,
                           unsigned long long* out2) {
    if (data == nullptr || out2 == nullptr || N < 0 || G < 0 || (layout != 0 && layout != 1)) return DSQ_ERR_ARG;
    dsq_pc::Digest d;
    if (elem_type == 0) d = dsq_pc::digest_host((const int32_t*)data, layout, N, G, n_threads);
    else if (elem_type == 1) d = dsq_pc::digest_host((const int64_t*)data, layout, N, G, n_threads);
    else if (elem_type == 2) d = dsq_pc::digest_host((const double*)data, layout, N, G, n_threads);
    else return DSQ_ERR_ARG;
    out2[0] = d.a;
    out2[1] = d.b;
    return DSQ_OK;
}

int dsq_plugin_cache_config(dsq_ctx* ctx, int enabled, long long budget_bytes) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (enabled >= 0) c.enabled = enabled != 0;
    if (budget_bytes >= 0) c.budget = (size_t)budget_bytes;
    return DSQ_OK;
}

int dsq_plugin_cache_clear(dsq_ctx* ctx) {
    DSQ_HIP(hipSetDevice(ctx->device));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->pc) dsq_pc::clear(*ctx->pc);
    return DSQ_OK;
}

int dsq_plugin_cache_stats(dsq_ctx* ctx, double* out, int n) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    const double v[13] = {(double)c.st.hits, (double)c.st.misses, (double)c.st.adopted, (double)c.st.evictions,
                          (double)c.st.h2d_bytes, (double)c.st.d2h_bytes, c.st.hash_ms, (double)c.resident,
                          (double)c.pooled, (double)c.ents.size(), (double)c.st.mallocs, (double)c.budget,
                          (double)c.st.verified};
    for (int i = 0; i < n && i < 13; ++i) out[i] = v[i];
    return DSQ_OK;
}

// ------------------------------------------------------------------ grid searches + trend GLM as entry points
int dsq_inf_grid_fit_alpha(dsq_ctx* ctx, const void* counts, int count_type, int count_layout, const double* design,
                           const double* mu, int mu_layout, int N, int G, int P, double min_disp, double max_disp,
                           double* log_alpha_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    const double* d_mu;
    PluginDesign::One* D;
    PcBuf a, work;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, true, &d_mu))) return rc;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    DSQ_HIP(a.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(work.alloc(ctx, (size_t)G * dsq::kAlphaGridWorkDoubles * sizeof(double)));
    DSQ_HIP(ensure_list(ctx, (size_t)G));
    std::vector<int32_t> all((size_t)G);
    for (int g = 0; g < G; ++g) all[(size_t)g] = g;
    DSQ_HIP(hipMemcpyAsync(ctx->d_list, all.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    // the production fallback path: 100 wavefronts per gene and level (k_alpha_grid_eval)
    DSQ_HIP(dsq::launch_alpha_grid(ctx->stream, d_y, d_mu, ldn, D->Xt, D->ldx, N, P, min_disp, max_disp, a.as<double>(),
                                   ctx->d_list, G, work.as<double>()));
    DSQ_HIP(hipMemcpyAsync(log_alpha_out, a.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    // the production kernel stores alpha = exp(best grid point), which is what fit_alpha_mle needs
    // (utils.py:557); grid_fit_alpha itself returns the grid point (grid_search.py:141-142): back to the log
    // (exp/log round trip: a few 1e-16 absolute)
    for (int g = 0; g < G; ++g) log_alpha_out[g] = std::log(log_alpha_out[g]);
    return DSQ_OK;
}

int dsq_inf_grid_fit_beta(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                          const double* size_factors, const double* design, const double* disp, int N, int G,
                          double min_mu, int grid_length, double min_beta, double max_beta, double* beta_out) {
    if (G <= 0) return DSQ_OK;
    DSQ_CHECK_ARG(grid_length >= 2, "grid_length must be at least 2");
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, d, b;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, 2, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    DSQ_HIP(b.alloc(ctx, (size_t)G * 2 * sizeof(double)));
    DSQ_HIP(dsq::launch_grid_beta(ctx->stream, d_y, ldn, sf.as<double>(), D->Xt, D->ldx, N, G, d.as<double>(), min_mu,
                                  min_beta, max_beta, grid_length, b.as<double>()));
    DSQ_HIP(hipMemcpyAsync(beta_out, b.p, (size_t)G * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_dispersion_trend_gamma_glm(dsq_ctx* ctx, const double* covariates, const double* targets, int n,
                                       double* coeffs2, double* predictions, int* converged) {
    DSQ_CHECK_ARG(n >= 1, "no genes");
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    PcBuf cov, tgt, keep;
    if ((rc = pc_upload_small(ctx, covariates, (size_t)n * sizeof(double), cov))) return rc;
    if ((rc = pc_upload_small(ctx, targets, (size_t)n * sizeof(double), tgt))) return rc;
    DSQ_HIP(keep.alloc(ctx, (size_t)n));
    double* d_out = ctx->d_scratch + 1536;
    DSQ_HIP(dsq::launch_trend_glm(ctx->stream, tgt.as<double>(), cov.as<double>(), n, keep.as<uint8_t>(), d_out));
    double out5[5];
    DSQ_HIP(hipMemcpyAsync(out5, d_out, sizeof(out5), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    coeffs2[0] = out5[0]; coeffs2[1] = out5[1];
    if (converged) *converged = (int)out5[2];
    if (predictions)  // covariates @ coeffs (default_inference.py:227)
        for (int i = 0; i < n; ++i) predictions[i] = out5[0] + covariates[i] * out5[1];
    return DSQ_OK;
}

// ------------------------------------------------------------------ the eight Inference methods
int dsq_inf_lin_reg_mu(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* size_factors, const double* design, int N, int G, int P, double min_mu,
                       double* mu_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, mu;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    DSQ_HIP(mu.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(dsq::launch_lin_mu(ctx->stream, d_y, ldn, sf.as<double>(), D->Xt, D->pinv, D->ldx, N, G, P, min_mu,
                               mu.as<double>()));
    if ((rc = pc_download_rows(ctx, mu_out, mu.as<double>(), ldn, N, G))) return rc;
    if ((rc = pc_adopt_f64(ctx, mu, N, G, ldn))) return rc;  // alpha_mle takes this matrix back (dds.py:778-785, 901-911)
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_irls2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                  const double* size_factors, const double* design, const double* disp, int N, int G, int P,
                  double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                  double* beta_out, double* mu_out, double* hat_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, d, beta, mu, hat, conv;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    // mixed designs: the kernel family of the pipeline (csrc/dsq_mix.h)
    if ((rc = pc_design(ctx, design, N, P, true, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    DSQ_HIP(beta.alloc(ctx, (size_t)G * P * sizeof(double)));
    DSQ_HIP(mu.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(hat.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    dsq::IrlsExtras exi{};
    if (D->mix != nullptr) exi.mix = &D->mix->d;
    rc = run_irls(ctx, d_y, ldn, sf.as<double>(), D->Xt, D->pinv, D->ldx, N, G, P, D->full_rank, d.as<double>(), min_mu,
                  beta_tol, min_beta, max_beta, maxiter, beta.as<double>(), mu.as<double>(), hat.as<double>(),
                  conv.as<uint8_t>(), nullptr, D->mix != nullptr ? &exi : nullptr, optimizer);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(beta_out, beta.p, (size_t)G * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = pc_download_rows(ctx, mu_out, mu.as<double>(), ldn, N, G))) return rc;
    if ((rc = pc_download_rows(ctx, hat_out, hat.as<double>(), ldn, N, G))) return rc;
    // mu comes back as mu_hat of alpha_mle (dds.py:757-785) or as mu of wald_test (ds.py:338-350)
    if ((rc = pc_adopt_f64(ctx, mu, N, G, ldn))) return rc;
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_irls(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                 const double* size_factors, const double* design, const double* disp, int N, int G, int P,
                 double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                 double* beta_out, double* mu_out, double* hat_out, uint8_t* converged) {
    return dsq_inf_irls2(ctx, counts, count_type, count_layout, size_factors, design, disp, N, G, P, min_mu, beta_tol,
                         min_beta, max_beta, maxiter, beta_out, mu_out, hat_out, converged, 0);
}

int dsq_inf_alpha_mle2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* design, const double* mu, int mu_layout, const double* alpha_hat, int N,
                       int G, int P, double min_disp, double max_disp, double prior_disp_var, int cr_reg,
                       int prior_reg, double* alpha_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    const double* d_mu;
    dsq_pc::Entry* ye = nullptr;
    PluginDesign::One* D;
    PcBuf ah, a, conv;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, true, &d_mu))) return rc;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y, &ye))) return rc;
    // mixed designs (categorical columns + up to three continuous covariates, csrc/dsq_mix.h): the kernel family of
    // the pipeline, here with mu gathered from the caller's matrix; genes with a count beyond its 16-bit staging stay on
    // the general kernel
    const bool want_mix = cr_reg != 0 && optimizer == 0;
    if ((rc = pc_design(ctx, design, N, P, want_mix, &D))) return rc;
    if ((rc = pc_upload_small(ctx, alpha_hat, (size_t)G * sizeof(double), ah))) return rc;
    DSQ_HIP(a.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    dsq::AlphaExtras ex{};
    const bool mix = want_mix && D->mix != nullptr;
    if (mix) {
        if ((rc = pc_mix_lists(ctx, ye, N, G, ldn))) return rc;
        ex.mix = &D->mix->d;
        ex.rows = (const int32_t*)ye->d_lists; ex.n_rows = ye->n_rows;
        ex.waves = (const int32_t*)ye->d_lists + ye->n_rows; ex.n_waves = ye->n_waves;
    }
    if ((rc = run_alpha(ctx, d_y, d_mu, ldn, D->Xt, D->ldx, N, G, P, ah.as<double>(), min_disp, max_disp,
                        prior_disp_var, cr_reg, prior_reg, a.as<double>(), conv.as<uint8_t>(), nullptr, nullptr,
                        DSQ_CONST_COMPUTE, mix ? &ex : nullptr, optimizer)))
        return rc;
    DSQ_HIP(hipMemcpyAsync(alpha_out, a.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_alpha_mle(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                      const double* design, const double* mu, int mu_layout, const double* alpha_hat, int N,
                      int G, int P, double min_disp, double max_disp, double prior_disp_var, int cr_reg,
                      int prior_reg, double* alpha_out, uint8_t* converged) {
    return dsq_inf_alpha_mle2(ctx, counts, count_type, count_layout, design, mu, mu_layout, alpha_hat, N, G, P, min_disp,
                              max_disp, prior_disp_var, cr_reg, prior_reg, alpha_out, converged, 0);
}

int dsq_inf_lfc_shrink_nbinom_glm2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                   const double* design, const double* size, const double* offset, int N, int G,
                                   int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                   double* beta_out, double* inv_hessian_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_SHRINK_MAX_P, "P out of range (apeGLM shrinkage: at most 48 design columns)");
    DSQ_CHECK_ARG(shrink_index >= 0 && shrink_index < P, "shrink_index out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sz, off, b, ih, conv;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size, (size_t)G * sizeof(double), sz))) return rc;
    if ((rc = pc_upload_small(ctx, offset, (size_t)N * sizeof(double), off))) return rc;
    DSQ_HIP(b.alloc(ctx, (size_t)G * P * sizeof(double)));
    DSQ_HIP(ih.alloc(ctx, (size_t)G * P * P * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    if ((rc = dsq_dev_lfc_shrink3(ctx, d_y, ldn, off.as<double>(), D->Xt, D->ldx, N, G, P, sz.as<double>(),
                                  prior_no_shrink_scale, prior_scale, shrink_index, b.as<double>(), ih.as<double>(),
                                  conv.as<uint8_t>(), nullptr, optimizer)))
        return rc;
    DSQ_HIP(hipMemcpyAsync(beta_out, b.p, (size_t)G * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(inv_hessian_out, ih.p, (size_t)G * P * P * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_lfc_shrink_nbinom_glm(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                  const double* design, const double* size, const double* offset, int N, int G,
                                  int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                  double* beta_out, double* inv_hessian_out, uint8_t* converged) {
    return dsq_inf_lfc_shrink_nbinom_glm2(ctx, counts, count_type, count_layout, design, size, offset, N, G, P,
                                          prior_no_shrink_scale, prior_scale, shrink_index, beta_out, inv_hessian_out,
                                          converged, 0);
}

int dsq_inf_wald_test(dsq_ctx* ctx, const double* design, const double* disp, const double* lfc,
                      const double* mu, int mu_layout, const double* ridge, const double* contrast,
                      double lfc_null, int alt, int N, int G, int P, double* pvals, double* stats,
                      double* se) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(alt >= 0 && alt <= 4, "unknown alternative hypothesis");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_mu;
    PluginDesign::One* D;
    PcBuf d, b, o;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, false, &d_mu))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    if ((rc = pc_upload_small(ctx, lfc, (size_t)G * P * sizeof(double), b))) return rc;
    DSQ_HIP(o.alloc(ctx, (size_t)3 * G * sizeof(double)));
    double* dp = o.as<double>();
    rc = dsq_dev_wald(ctx, d_mu, ldn, nullptr, D->Xt, D->ldx, N, G, P, d.as<double>(), b.as<double>(), ridge, contrast,
                      lfc_null, alt, dp, dp + G, dp + 2 * (size_t)G);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(pvals, dp, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(stats, dp + G, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(se, dp + 2 * (size_t)G, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_fit_rough_dispersions(dsq_ctx* ctx, const double* normed, int layout, const double* design,
                                  int N, int G, int P, double* alpha_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion. Please use a design with "
                          "fewer variables.");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_v;
    PluginDesign::One* D;
    PcBuf o;
    if ((rc = pc_f64(ctx, normed, layout, N, G, ldn, false, &d_v))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    DSQ_HIP(o.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(dsq::launch_rough_from_normed(ctx->stream, d_v, ldn, D->Xt, D->pinv, D->ldx, N, G, P, o.as<double>()));
    DSQ_HIP(hipMemcpyAsync(alpha_out, o.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

// all_zero (nullable): per gene, 1 when every normalised count of the gene is zero - the reference drops those columns
// before it takes the moments (utils.py:878), so the caller drops the same entries of alpha_out
int dsq_inf_fit_moments_dispersions2(dsq_ctx* ctx, const double* normed, int layout, const double* size_factors,
                                     int N, int G, double* alpha_out, uint8_t* all_zero) {
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_v;
    PcBuf o, fl;
    if ((rc = pc_f64(ctx, normed, layout, N, G, ldn, false, &d_v))) return rc;
    double smi = 0.0;
    for (int n = 0; n < N; ++n) smi += 1.0 / size_factors[n];
    smi /= (double)N;
    DSQ_HIP(o.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(dsq::launch_moments_from_normed(ctx->stream, d_v, ldn, N, G, smi, o.as<double>()));
    DSQ_HIP(hipMemcpyAsync(alpha_out, o.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (all_zero != nullptr) {
        DSQ_HIP(fl.alloc(ctx, (size_t)G));
        hipLaunchKernelGGL(dsq_pc::k_rows_all_zero, dim3((G + 3) / 4), dim3(256), 0, ctx->stream, d_v, ldn, N, G,
                           fl.as<uint8_t>());
        DSQ_HIP(hipGetLastError());
        DSQ_HIP(hipMemcpyAsync(all_zero, fl.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_fit_moments_dispersions(dsq_ctx* ctx, const double* normed, int layout,
                                    const double* size_factors, int N, int G, double* alpha_out) {
    return dsq_inf_fit_moments_dispersions2(ctx, normed, layout, size_factors, N, G, alpha_out, nullptr);
}

}  // extern "C"
