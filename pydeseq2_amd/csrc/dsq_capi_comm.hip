// dsq_capi_comm.hip — RCCL exchanges (dlopen) and the sample-sharded size-factor steps of the gene-sharded pipeline.
#include "dsq_capi_internal.h"

// ------------------------------------------------------------------ RCCL (multi-GPU exchanges)
// librccl is resolved lazily with dlopen so that single-GPU use does not depend on it.
struct Uid { char internal[128]; };
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /*ncclUniqueId by value: 128 bytes*/ Uid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool load_rccl(std::string& err) {
    if (g_rccl.h) return true;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("dlopen(librccl.so): ") + dlerror(); return false; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather) {
        err = "librccl.so lacks the expected nccl* symbols";
        return false;
    }
    g_rccl.h = h;
    return true;
}
#define DSQ_NCCL(call)                                                                             \
    do {                                                                                           \
        int r_ = (call);                                                                           \
        if (r_ != 0)                                                                               \
            return fail(ctx, DSQ_ERR_HIP, std::string(#call) + ": " +                              \
                                              (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); \
    } while (0)
}  // namespace

extern "C" {

int dsq_comm_unique_id(dsq_ctx* ctx, char* out128, int len) {
    DSQ_CHECK_ARG(len >= 128, "unique id buffer must hold 128 bytes");
    if (!load_rccl(ctx->err)) return DSQ_ERR_HIP;
    Uid id;
    DSQ_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, id.internal, 128);
    return DSQ_OK;
}

int dsq_comm_init(dsq_ctx* ctx, const char* uid128, int rank, int world) {
    DSQ_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
    if (!load_rccl(ctx->err)) return DSQ_ERR_HIP;
    DSQ_HIP(hipSetDevice(ctx->device));
    Uid id;
    memcpy(id.internal, uid128, 128);
    DSQ_NCCL(g_rccl.CommInitRank(&ctx->comm, world, id, rank));
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return DSQ_OK;
}

int dsq_comm_destroy(dsq_ctx* ctx) {
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return DSQ_OK;
}

// in-place all-reduce (sum) on a device buffer; dtype: 0 = uint32, 1 = float64
int dsq_comm_allreduce_sum(dsq_ctx* ctx, void* d_buf, size_t count, int dtype) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    const int nccl_type = dtype == 0 ? 3 /*ncclUint32*/ : 8 /*ncclFloat64*/;
    DSQ_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, nccl_type, 0 /*ncclSum*/, ctx->comm, ctx->stream));
    return DSQ_OK;
}

// all-gather of `bytes_per_rank` bytes from every rank into d_recv (world * bytes_per_rank)
int dsq_comm_allgather(dsq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    DSQ_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, 0 /*ncclInt8*/, ctx->comm, ctx->stream));
    return DSQ_OK;
}

// what the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank): bench.py prints it in the result line
int dsq_comm_info(dsq_ctx* ctx, int* nranks, int* rank) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    DSQ_CHECK_ARG(g_rccl.CommCount != nullptr && g_rccl.CommUserRank != nullptr, "librccl.so lacks ncclCommCount");
    if (nranks) DSQ_NCCL(g_rccl.CommCount(ctx->comm, nranks));
    if (rank) DSQ_NCCL(g_rccl.CommUserRank(ctx->comm, rank));
    return DSQ_OK;
}


// gene-sharded trend exchange (pydeseq2_amd/distributed.py): both per-gene vectors of a rank in ONE send buffer, NaN-padded
// to `len` genes each; after the all-gather the [world][2][len] block is split into the two [world * len] vectors the
// trend / prior kernels read
int dsq_dev_pack2(dsq_ctx* ctx, const double* d_a, const double* d_b, int n, int len, double* d_send) {
    DSQ_CHECK_ARG(n >= 0 && n <= len, "n out of range");
    DSQ_HIP(dsq::launch_pack2(ctx->stream, d_a, d_b, n, len, d_send));
    return DSQ_OK;
}
int dsq_dev_unzip2(dsq_ctx* ctx, const double* d_recv, int world, int len, double* d_a_all, double* d_b_all) {
    DSQ_HIP(dsq::launch_unzip2(ctx->stream, d_recv, world, len, d_a_all, d_b_all));
    return DSQ_OK;
}

// ---- size factors, one pass at a time (distributed median of ratios)
int dsq_dev_sf_keys(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G, const double* d_logmeans,
                    const uint8_t* d_gene_mask, void* d_keys) {
    DSQ_HIP(dsq::launch_sf_keys(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_gene_mask,
                                (unsigned long long*)d_keys));
    return DSQ_OK;
}
int dsq_dev_sf_keys_compact(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                            const double* d_logmeans, const uint8_t* d_gene_mask, int32_t* d_idx_work, void* d_keys,
                            int* h_n_usable) {
    DSQ_HIP(dsq::launch_sf_compact(ctx->stream, d_logmeans, d_gene_mask, G, d_idx_work));
    int gu = 0;
    DSQ_HIP(hipMemcpyAsync(&gu, d_idx_work + G, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(dsq::launch_sf_keys_compact(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_idx_work,
                                        (unsigned long long*)d_keys));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    *h_n_usable = gu;
    return DSQ_OK;
}
int dsq_dev_sf_count(dsq_ctx* ctx, const void* d_keys, int N, int G, uint32_t* d_counts) {
    DSQ_HIP(dsq::launch_sf_count(ctx->stream, (const unsigned long long*)d_keys, N, G, d_counts));
    return DSQ_OK;
}
int dsq_dev_sf_init(dsq_ctx* ctx, const uint32_t* d_total, int N, void* d_prefix, uint32_t* d_rank) {
    DSQ_HIP(dsq::launch_sf_init(ctx->stream, d_total, N, (unsigned long long*)d_prefix, d_rank));
    return DSQ_OK;
}
int dsq_dev_sf_hist(dsq_ctx* ctx, const void* d_keys, int N, int G, const void* d_prefix, int shift,
                    uint32_t* d_hist) {
    DSQ_HIP(dsq::launch_sf_hist(ctx->stream, (const unsigned long long*)d_keys, N, G,
                                (const unsigned long long*)d_prefix, shift, d_hist));
    return DSQ_OK;
}
int dsq_dev_sf_pick(dsq_ctx* ctx, const uint32_t* d_hist, int N, int shift, void* d_prefix, uint32_t* d_rank) {
    DSQ_HIP(dsq::launch_sf_pick(ctx->stream, d_hist, N, shift, (unsigned long long*)d_prefix, d_rank));
    return DSQ_OK;
}
int dsq_dev_sf_finish(dsq_ctx* ctx, const void* d_prefix, const uint32_t* d_total, int N, double* d_sf) {
    DSQ_HIP(dsq::launch_sf_finish(ctx->stream, (const unsigned long long*)d_prefix, d_total, N, d_sf));
    return DSQ_OK;
}

}  // extern "C"

