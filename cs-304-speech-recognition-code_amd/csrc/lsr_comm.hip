// The two exchange steps of the multi-GPU path, on RCCL directly (one process per GPU, the
// communicator bootstrapped from an ncclUniqueId the caller distributes):
//   lsr_allgather_labels  every rank ends with the label records of all utterances in global order,
//   lsr_allreduce_stats   the packed float64 training statistics are summed over the ranks.
// Both move one fused buffer per call on the context's stream.
//
// Utterances are sharded in contiguous ranges (lsr_shard_bounds: sizes differ by at most one, earlier
// ranks take the extra item).  A label record is label_cap word indices followed by the int32 label
// count, little endian; every rank contributes shard_cap = size of the largest shard records, unused
// ones filled with 0xFF, so that a single ncclAllGather of equal pieces suffices.
//
// RCCL is loaded with dlopen on first use (librccl.so.1; a copy already loaded by the host program,
// e.g. by PyTorch, is reused), so loading liblsr_hip.so itself does not pull it in.
#include <dlfcn.h>
#include <string.h>

#include "lsr_internal.h"

// the subset of the RCCL API used here (types as in rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;      // ncclSuccess = 0
enum { LSR_NCCL_UINT8 = 1, LSR_NCCL_FLOAT64 = 8, LSR_NCCL_SUM = 0 };      // ncclUint8, ncclFloat64, ncclSum

struct lsr_rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static lsr_rccl g_rccl;

static int rccl_load(lsr_ctx* ctx) {
    if (g_rccl.handle) return LSR_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return lsr_fail(ctx, LSR_ERR_STATE, "RCCL not found: %s", dlerror());
    lsr_rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString)
        return lsr_fail(ctx, LSR_ERR_STATE, "RCCL library lacks a required symbol");
    g_rccl = r;
    return LSR_OK;
}

#define LSR_NCCL_CHECK(ctx, expr)                                                                      \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != 0)                                                                                   \
            return lsr_fail((ctx), LSR_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));   \
    } while (0)

struct lsr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    lsr_buffer send, recv;
};

// ---- sharding and the record layout (host functions, usable without a GPU) -----------------------------
extern "C" void lsr_shard_bounds(int64_t n_items, int rank, int world, int64_t* lo, int64_t* hi) {
    const int64_t base = n_items / world, extra = n_items % world;
    const int64_t l = rank * base + (rank < extra ? rank : extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < extra ? 1 : 0);
}

static int64_t shard_cap_of(int64_t n_total, int world) {
    int64_t lo, hi;
    lsr_shard_bounds(n_total, 0, world, &lo, &hi);
    return hi - lo;
}

extern "C" int64_t lsr_label_record_bytes(int label_cap) { return (int64_t)label_cap + 4; }

extern "C" int lsr_host_pack_label_records(const uint8_t* labels, const int32_t* n_labels, int64_t n_local, int label_cap,
                                           int64_t shard_cap, uint8_t* records) {
    if (!labels || !n_labels || !records || n_local < 0 || n_local > shard_cap || label_cap < 1) return LSR_ERR_ARG;
    const int64_t rb = lsr_label_record_bytes(label_cap);
    memset(records, 0xFF, (size_t)(shard_cap * rb));
    for (int64_t i = 0; i < n_local; ++i) {
        memcpy(records + i * rb, labels + i * label_cap, (size_t)label_cap);
        const uint32_t c = (uint32_t)n_labels[i];
        for (int k = 0; k < 4; ++k) records[i * rb + label_cap + k] = (uint8_t)(c >> (8 * k));
    }
    return LSR_OK;
}

extern "C" int lsr_host_unpack_label_records(const uint8_t* records, int world, int64_t shard_cap, int64_t n_total,
                                             int label_cap, uint8_t* labels, int32_t* n_labels) {
    if (!records || !labels || !n_labels || world < 1 || label_cap < 1 || n_total < 0) return LSR_ERR_ARG;
    if (shard_cap != shard_cap_of(n_total, world)) return LSR_ERR_ARG;
    const int64_t rb = lsr_label_record_bytes(label_cap);
    for (int r = 0; r < world; ++r) {
        int64_t lo, hi;
        lsr_shard_bounds(n_total, r, world, &lo, &hi);
        for (int64_t g = lo; g < hi; ++g) {
            const uint8_t* rec = records + ((int64_t)r * shard_cap + (g - lo)) * rb;
            memcpy(labels + g * label_cap, rec, (size_t)label_cap);
            uint32_t c = 0;
            for (int k = 0; k < 4; ++k) c |= (uint32_t)rec[label_cap + k] << (8 * k);
            n_labels[g] = (int32_t)c;
        }
    }
    return LSR_OK;
}

// ---- device side of the same layout ---------------------------------------------------------------------
__global__ void lsr_pack_records_kernel(const uint8_t* __restrict__ labels, const int32_t* __restrict__ n_labels,
                                        int64_t n_local, int label_cap, int64_t shard_cap, uint8_t* __restrict__ records) {
    const int64_t rb = (int64_t)label_cap + 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one thread per record byte
    if (e >= shard_cap * rb) return;
    const int64_t i = e / rb;
    const int k = (int)(e - i * rb);
    uint8_t v = 0xFF;
    if (i < n_local) v = k < label_cap ? labels[i * label_cap + k] : (uint8_t)((uint32_t)n_labels[i] >> (8 * (k - label_cap)));
    records[e] = v;
}

__global__ void lsr_unpack_records_kernel(const uint8_t* __restrict__ records, int world, int64_t shard_cap, int64_t n_total,
                                          int label_cap, uint8_t* __restrict__ labels, int32_t* __restrict__ n_labels) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one thread per utterance of the job
    if (g >= n_total) return;
    const int64_t base = n_total / world, extra = n_total % world;
    // rank that owns utterance g: the first `extra` shards hold base + 1 utterances
    int64_t r, lo;
    if (g < extra * (base + 1)) {
        r = g / (base + 1);
        lo = r * (base + 1);
    } else {
        r = extra + (base > 0 ? (g - extra * (base + 1)) / base : 0);
        lo = extra * (base + 1) + (r - extra) * base;
    }
    const int64_t rb = (int64_t)label_cap + 4;
    const uint8_t* rec = records + (r * shard_cap + (g - lo)) * rb;
    for (int k = 0; k < label_cap; ++k) labels[g * label_cap + k] = rec[k];
    uint32_t c = 0;
    for (int k = 0; k < 4; ++k) c |= (uint32_t)rec[label_cap + k] << (8 * k);
    n_labels[g] = (int32_t)c;
}

extern "C" int lsr_pack_label_records(lsr_ctx* ctx, const uint8_t* labels_d, const int32_t* n_labels_d, int64_t n_local,
                                      int label_cap, int64_t shard_cap, uint8_t* records_d) {
    if (!ctx) return LSR_ERR_ARG;
    // a rank without utterances (more ranks than utterances) has nothing to point at: its records are all padding
    if (((!labels_d || !n_labels_d) && n_local != 0) || !records_d || label_cap < 1 || n_local < 0 || n_local > shard_cap || shard_cap < 1)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_pack_label_records: bad argument");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t piece = (size_t)(shard_cap * lsr_label_record_bytes(label_cap));
    hipLaunchKernelGGL(lsr_pack_records_kernel, dim3((unsigned)((piece + 255) / 256)), dim3(256), 0, ctx->stream, labels_d,
                       n_labels_d, n_local, label_cap, shard_cap, records_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_pack_records_kernel");
    return LSR_OK;
}

extern "C" int lsr_unpack_label_records(lsr_ctx* ctx, const uint8_t* records_d, int world, int64_t n_total, int label_cap,
                                        uint8_t* labels_out_d, int32_t* n_labels_out_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!records_d || !labels_out_d || !n_labels_out_d || world < 1 || label_cap < 1 || n_total < 1)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_unpack_label_records: bad argument");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(lsr_unpack_records_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, ctx->stream,
                       records_d, world, shard_cap_of(n_total, world), n_total, label_cap, labels_out_d, n_labels_out_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_unpack_records_kernel");
    return LSR_OK;
}

// ---- communicator ------------------------------------------------------------------------------------------
extern "C" int lsr_comm_get_unique_id(lsr_ctx* ctx, void* id_out) {
    if (!ctx || !id_out) return LSR_ERR_ARG;
    int rc = rccl_load(ctx);
    if (rc) return rc;
    ncclUniqueId id;
    LSR_NCCL_CHECK(ctx, g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, sizeof(id.internal));
    return LSR_OK;
}

extern "C" int lsr_comm_destroy(lsr_ctx* ctx) {
    if (!ctx) return LSR_ERR_ARG;
    lsr_comm* c = ctx->comm;
    if (!c) return LSR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->send.ptr) (void)hipFree(c->send.ptr);
    if (c->recv.ptr) (void)hipFree(c->recv.ptr);
    delete c;
    ctx->comm = nullptr;
    return LSR_OK;
}

extern "C" int lsr_comm_init(lsr_ctx* ctx, const void* unique_id, int rank, int world) {
    if (!ctx || !unique_id) return LSR_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_comm_init: rank %d of %d", rank, world);
    int rc = rccl_load(ctx);
    if (rc) return rc;
    if (ctx->comm) lsr_comm_destroy(ctx);
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, unique_id, sizeof(id.internal));
    lsr_comm* c = new lsr_comm();
    c->rank = rank;
    c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        delete c;
        return lsr_fail(ctx, LSR_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    ctx->comm = c;
    return LSR_OK;
}

extern "C" int lsr_allgather_labels(lsr_ctx* ctx, const uint8_t* labels_d, const int32_t* n_labels_d, int64_t n_local,
                                    int label_cap, int64_t n_total, uint8_t* labels_out_d, int32_t* n_labels_out_d) {
    if (!ctx) return LSR_ERR_ARG;
    lsr_comm* c = ctx->comm;
    if (!c) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_allgather_labels: call lsr_comm_init first");
    if (((!labels_d || !n_labels_d) && n_local != 0) || !labels_out_d || !n_labels_out_d || label_cap < 1 || n_total < 1)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_allgather_labels: bad argument");
    int64_t lo, hi;
    lsr_shard_bounds(n_total, c->rank, c->world, &lo, &hi);
    if (n_local != hi - lo)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_allgather_labels: rank %d of %d holds %lld utterances, its shard of %lld has %lld",
                        c->rank, c->world, (long long)n_local, (long long)n_total, (long long)(hi - lo));
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t cap = shard_cap_of(n_total, c->world), rb = lsr_label_record_bytes(label_cap);
    const size_t piece = (size_t)(cap * rb);
    int rc;
    if ((rc = lsr_reserve(ctx, &c->send, piece, "label records (send)"))) return rc;
    if ((rc = lsr_reserve(ctx, &c->recv, piece * c->world, "label records (receive)"))) return rc;
    if ((rc = lsr_pack_label_records(ctx, labels_d, n_labels_d, n_local, label_cap, cap, (uint8_t*)c->send.ptr))) return rc;
    LSR_NCCL_CHECK(ctx, g_rccl.AllGather(c->send.ptr, c->recv.ptr, piece, LSR_NCCL_UINT8, c->comm, ctx->stream));
    return lsr_unpack_label_records(ctx, (const uint8_t*)c->recv.ptr, c->world, n_total, label_cap, labels_out_d,
                                    n_labels_out_d);
}

extern "C" int lsr_allreduce_stats(lsr_ctx* ctx, double* stats_d, int64_t count) {
    if (!ctx) return LSR_ERR_ARG;
    lsr_comm* c = ctx->comm;
    if (!c) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_allreduce_stats: call lsr_comm_init first");
    if (!stats_d || count < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_allreduce_stats: bad argument");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    LSR_NCCL_CHECK(ctx, g_rccl.AllReduce(stats_d, stats_d, (size_t)count, LSR_NCCL_FLOAT64, LSR_NCCL_SUM, c->comm, ctx->stream));
    return LSR_OK;
}
