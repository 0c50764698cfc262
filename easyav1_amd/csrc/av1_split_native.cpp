// The tile-column split's per-frame driver and its collective, natively (include/easyav1_amd_split.h): reconstruction of the rank's strip,
// all-gather of the edge columns, in-loop filters, all-gather of the finished strips -- the sequence easyav1_amd/parallel.py drives from
// Python -- on one HIP stream, with RCCL (loaded at run time) or a caller's function as the all-gather.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ea1_common.h"
#include "easyav1_amd_split.h"

namespace {

constexpr uint32_t kHalo = 16;   // luma columns either side of a strip (easyav1_amd_video.h: why 16 suffice)

// ---- RCCL, bound at run time: the four calls and the two types the split needs (rccl.h: ncclUniqueId is 128 bytes, passed by value;
//      ncclUint8 = 1; ncclSuccess = 0).  One binding per process.
struct NcclUniqueId {
    char internal[128];
};
using NcclComm = void *;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) return;
        auto sym = [&](const char *n) { return dlsym(g_rccl.lib, n); };
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
        g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
            dlclose(g_rccl.lib);
            g_rccl.lib = nullptr;
        }
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}

int rccl_gather(void *userdata, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    const Rccl *r = rccl();
    if (!r) return -1;
    return r->AllGather(d_send, d_recv, bytes, 1 /* ncclUint8 */, static_cast<NcclComm>(userdata), static_cast<hipStream_t>(stream));
}

}  // namespace

struct ea1_split {
    ea1_recon *ctx = nullptr;
    int device = 0, world = 1, rank = 0, nchains = 1;
    std::vector<uint32_t> bounds;   // world + 1
    uint32_t max_w = 0;
    ea1_all_gather_fn gather = nullptr;
    void *userdata = nullptr;
    NcclComm comm = nullptr;        // the split's own communicator (RCCL form)
    // device buffers, per collective: [chain][bytes per chain] contributed, [rank][chain][bytes per chain] received
    size_t halo_chain = 0, strip_chain = 0;   // bytes per chain (rounded up to 16)
    uint8_t *halo_send = nullptr, *halo_recv = nullptr, *strip_send = nullptr, *strip_recv = nullptr;
};

static void free_buffers(ea1_split *s)
{
    for (uint8_t **p : {&s->halo_send, &s->halo_recv, &s->strip_send, &s->strip_recv}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

extern "C" int ea1_split_create(ea1_split **out, ea1_recon *ctx, const uint32_t *x_bounds, int world, int rank, ea1_all_gather_fn gather, void *userdata)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_split_create: out is NULL");
    *out = nullptr;
    if (!ctx || !x_bounds || world < 1 || world > 64 || rank < 0 || rank >= world || (world > 1 && !gather))
        return ea1::fail(EA1_EINVAL, "ea1_split_create: bad arguments (world %d, rank %d)", world, rank);
    for (int q = 0; q < world; q++)
        if (x_bounds[q] >= x_bounds[q + 1] || (x_bounds[q] & 63u) || (q + 1 < world && x_bounds[q + 1] - x_bounds[q] < 2 * kHalo))
            return ea1::fail(EA1_EINVAL, "ea1_split_create: strip %d = [%u, %u) (ascending multiples of 64, at least 32 columns each)", q, x_bounds[q], x_bounds[q + 1]);
    if (x_bounds[0] != 0) return ea1::fail(EA1_EINVAL, "ea1_split_create: the first strip starts at column %u", x_bounds[0]);
    if (ea1_recon_layout(ctx) != EA1_LAYOUT_420) return ea1::fail(EA1_ENOTSUP, "ea1_split_create: the tile-column split is 4:2:0 only");
    auto *s = new (std::nothrow) ea1_split();
    if (!s) return ea1::fail(EA1_ENOMEM, "out of memory");
    s->ctx = ctx;
    s->world = world, s->rank = rank;
    s->nchains = ea1_recon_num_chains(ctx);
    s->bounds.assign(x_bounds, x_bounds + world + 1);
    for (int q = 0; q < world; q++) s->max_w = std::max(s->max_w, x_bounds[q + 1] - x_bounds[q]);
    s->gather = gather, s->userdata = userdata;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;   // (the rank's device is the calling thread's current one: the context was made on it)
    if (world > 1 || gather) {
        auto up = [](size_t n) { return (n + 15) & ~size_t(15); };
        s->halo_chain = up(2 * ea1_recon_columns_bytes(ctx, kHalo));
        s->strip_chain = up(ea1_recon_columns_bytes(ctx, (s->max_w + 7u) & ~7u));
        const size_t nc = size_t(s->nchains);
        if (hipMalloc(reinterpret_cast<void **>(&s->halo_send), nc * s->halo_chain) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&s->halo_recv), nc * s->halo_chain * size_t(world)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&s->strip_send), nc * s->strip_chain) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&s->strip_recv), nc * s->strip_chain * size_t(world)) != hipSuccess) {
            free_buffers(s);
            delete s;
            return ea1::fail(EA1_ENOMEM, "ea1_split_create: out of device memory for the exchange buffers");
        }
        // (columns beyond a strip narrower than the widest one travel as they are: zero them once so that nothing uninitialised crosses the wire)
        (void)hipMemset(s->strip_send, 0, nc * s->strip_chain);
        (void)hipMemset(s->halo_send, 0, nc * s->halo_chain);
    }
    *out = s;
    return EA1_OK;
}

extern "C" int ea1_rccl_unique_id(void *id128)
{
    if (!id128) return ea1::fail(EA1_EINVAL, "ea1_rccl_unique_id: NULL argument");
    const Rccl *r = rccl();
    if (!r) return ea1::fail(EA1_ENODEV, "ea1_rccl_unique_id: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "not found");
    NcclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc != 0) return ea1::fail(EA1_EHIP, "ncclGetUniqueId: %s", r->GetErrorString ? r->GetErrorString(rc) : "failed");
    memcpy(id128, &id, sizeof id);
    return EA1_OK;
}

extern "C" int ea1_split_create_rccl(ea1_split **out, ea1_recon *ctx, const uint32_t *x_bounds, int world, int rank, const void *unique_id)
{
    if (!out) return ea1::fail(EA1_EINVAL, "ea1_split_create_rccl: out is NULL");
    *out = nullptr;
    if (!unique_id) return ea1::fail(EA1_EINVAL, "ea1_split_create_rccl: no unique id");
    const Rccl *r = rccl();
    if (!r) return ea1::fail(EA1_ENODEV, "ea1_split_create_rccl: librccl.so could not be loaded");
    NcclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    NcclComm comm = nullptr;
    const int nrc = r->CommInitRank(&comm, world, id, rank);
    if (nrc != 0 || !comm) return ea1::fail(EA1_EHIP, "ncclCommInitRank (rank %d of %d): %s", rank, world, r->GetErrorString ? r->GetErrorString(nrc) : "failed");
    const int rc = ea1_split_create(out, ctx, x_bounds, world, rank, rccl_gather, comm);
    if (rc != EA1_OK) {
        (void)r->CommDestroy(comm);
        return rc;
    }
    (*out)->comm = comm;
    return EA1_OK;
}

extern "C" int ea1_split_job_create(ea1_split *s, const ea1_ir_frame *f, ea1_job **out)
{
    if (!s || !f || !out) return ea1::fail(EA1_EINVAL, "ea1_split_job_create: NULL argument");
    const ea1_lr_unit *lr[3] = {f->lr[0], f->lr[1], f->lr[2]};
    return ea1_job_create_strip(s->ctx, &f->desc, f->blocks, f->txs, f->coefs, f->aux, f->cdef_idx, lr, s->bounds[size_t(s->rank)], s->bounds[size_t(s->rank) + 1], out);
}

extern "C" int ea1_split_decode(ea1_split *s, ea1_job *const *jobs, int dst_slot, void *stream)
{
    if (!s || !jobs) return ea1::fail(EA1_EINVAL, "ea1_split_decode: NULL argument");
    if (dst_slot < 0 || dst_slot >= EA1_NUM_REF_SLOTS) return ea1::fail(EA1_EINVAL, "ea1_split_decode: destination slot %d", dst_slot);
    const uint32_t x0 = s->bounds[size_t(s->rank)], x1 = s->bounds[size_t(s->rank) + 1];
    const size_t halo = ea1_recon_columns_bytes(s->ctx, kHalo);
    int rc = ea1_recon_submit_phase(s->ctx, jobs, EA1_PHASE_RECON, stream);
    if (rc != EA1_OK) return rc;
    const bool exchange = s->gather != nullptr;
    if (exchange) {
        // 1. the strip's own edge columns of the reconstruction: [left edge | right edge] per chain
        for (int c = 0; c < s->nchains && rc == EA1_OK; c++) {
            if (!jobs[c]) continue;
            uint8_t *mine = s->halo_send + size_t(c) * s->halo_chain;
            rc = ea1_recon_get_columns(s->ctx, c, EA1_PIC_WORK, x0, kHalo, mine, stream);
            if (rc == EA1_OK) rc = ea1_recon_get_columns(s->ctx, c, EA1_PIC_WORK, x1 - kHalo, kHalo, mine + halo, stream);
        }
        if (rc != EA1_OK) return rc;
        const size_t per_rank = size_t(s->nchains) * s->halo_chain;
        if (s->gather(s->userdata, s->halo_send, s->halo_recv, per_rank, stream) != 0) return ea1::fail(EA1_EHIP, "ea1_split_decode: the all-gather of the edge columns failed");
        // 2. the neighbours' edges beside the strip: the left neighbour's RIGHT edge, the right neighbour's LEFT edge
        for (int c = 0; c < s->nchains && rc == EA1_OK; c++) {
            if (!jobs[c]) continue;
            if (s->rank > 0)
                rc = ea1_recon_put_columns(s->ctx, c, EA1_PIC_WORK, x0 - kHalo, kHalo, s->halo_recv + size_t(s->rank - 1) * per_rank + size_t(c) * s->halo_chain + halo, stream);
            if (rc == EA1_OK && s->rank + 1 < s->world)
                rc = ea1_recon_put_columns(s->ctx, c, EA1_PIC_WORK, x1, kHalo, s->halo_recv + size_t(s->rank + 1) * per_rank + size_t(c) * s->halo_chain, stream);
        }
        if (rc != EA1_OK) return rc;
    }
    rc = ea1_recon_submit_phase(s->ctx, jobs, EA1_PHASE_FILTER, stream);
    if (rc != EA1_OK || !exchange) return rc;
    // 3. the finished strips of everybody into everybody's reference slot
    for (int c = 0; c < s->nchains && rc == EA1_OK; c++)
        if (jobs[c]) rc = ea1_recon_get_columns(s->ctx, c, dst_slot, x0, x1 - x0, s->strip_send + size_t(c) * s->strip_chain, stream);
    if (rc != EA1_OK) return rc;
    const size_t per_rank = size_t(s->nchains) * s->strip_chain;
    if (s->gather(s->userdata, s->strip_send, s->strip_recv, per_rank, stream) != 0) return ea1::fail(EA1_EHIP, "ea1_split_decode: the all-gather of the finished strips failed");
    for (int q = 0; q < s->world && rc == EA1_OK; q++) {
        if (q == s->rank) continue;
        const uint32_t a = s->bounds[size_t(q)], e = s->bounds[size_t(q) + 1];
        for (int c = 0; c < s->nchains && rc == EA1_OK; c++)
            if (jobs[c]) rc = ea1_recon_put_columns(s->ctx, c, dst_slot, a, e - a, s->strip_recv + size_t(q) * per_rank + size_t(c) * s->strip_chain, stream);
    }
    return rc;
}

extern "C" void ea1_split_traffic(const ea1_split *s, size_t *halo_sent, size_t *halo_received, size_t *strip_sent, size_t *strip_received)
{
    const size_t nc = s ? size_t(s->nchains) : 0, w = s ? size_t(s->world) : 0;
    if (halo_sent) *halo_sent = s ? nc * s->halo_chain : 0;
    if (halo_received) *halo_received = s ? nc * s->halo_chain * w : 0;
    if (strip_sent) *strip_sent = s ? nc * s->strip_chain : 0;
    if (strip_received) *strip_received = s ? nc * s->strip_chain * w : 0;
}

extern "C" void ea1_split_destroy(ea1_split *s)
{
    if (!s) return;
    (void)hipDeviceSynchronize();   // nothing of the split's buffers may still be in flight
    if (s->comm)
        if (const Rccl *r = rccl()) (void)r->CommDestroy(s->comm);
    free_buffers(s);
    delete s;
}
