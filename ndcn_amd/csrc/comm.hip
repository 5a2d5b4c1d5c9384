// Multi-GPU transport of the path behind the C ABI: RCCL over xGMI, one rank per GPU.
//
//   ndcn_comm        a communicator (ncclComm_t created here from a unique id, or adopted from the caller)
//   ndcn_halo_plan   what this rank sends / receives before each A X: per peer row counts + the rows of the own panel to
//                    pack.  ndcn_halo_exchange_f32 = gather kernel + ONE grouped ncclSend / ncclRecv per peer (an
//                    all-to-all-v: only the referenced rows travel, point to point over xGMI - not an all-gather of the panel)
//   allreduce        the 16 bytes per adaptive step the controller needs ({sum of squared error ratios, non-finite count})
//
// librccl is bound at first use with dlopen / dlsym (the process' already loaded librccl.so.1 when the caller - e.g.
// PyTorch - brought one): the library keeps loading on hosts without RCCL, and fails loudly when a multi-GPU entry point
// is called there.
#include <dlfcn.h>
#include <string.h>
#include <new>
#include <vector>

#include "kernels.h"

namespace {

typedef struct ncclComm *nccl_comm_t;
typedef struct { char internal[128]; } nccl_unique_id;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8 };

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl q;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            q.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (q.lib) break;
        }
        if (!q.lib) return q;
#define NDCN_SYM(field, sym) q.field = reinterpret_cast<decltype(q.field)>(dlsym(q.lib, sym))
        NDCN_SYM(GetUniqueId, "ncclGetUniqueId");
        NDCN_SYM(CommInitRank, "ncclCommInitRank");
        NDCN_SYM(CommDestroy, "ncclCommDestroy");
        NDCN_SYM(Send, "ncclSend");
        NDCN_SYM(Recv, "ncclRecv");
        NDCN_SYM(GroupStart, "ncclGroupStart");
        NDCN_SYM(GroupEnd, "ncclGroupEnd");
        NDCN_SYM(AllReduce, "ncclAllReduce");
        NDCN_SYM(GetErrorString, "ncclGetErrorString");
#undef NDCN_SYM
        q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.Send && q.Recv && q.GroupStart && q.GroupEnd && q.AllReduce;
        return q;
    }();
    return r;
}

int need_rccl(const char *who) {
    if (rccl().ok) return NDCN_OK;
    ndcn::set_error("%s: librccl.so.1 could not be loaded (%s) - the multi-GPU entry points need RCCL", who,
                    rccl().lib ? "symbols missing" : dlerror());
    return NDCN_EHIP;
}

#define NDCN_NCCL(call)                                                                                          \
    do {                                                                                                         \
        int r_ = (call);                                                                                         \
        if (r_ != kNcclSuccess) {                                                                                \
            ndcn::set_error("%s: %s failed: %s", __func__, #call, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
            return NDCN_EHIP;                                                                                    \
        }                                                                                                        \
    } while (0)

}  // namespace

struct ndcn_comm {
    nccl_comm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
};

struct ndcn_halo_plan {
    ndcn_comm *c = nullptr;
    int64_t n_halo = 0, n_send = 0;
    std::vector<int64_t> send_counts, recv_counts, send_off, recv_off;
    const int32_t *d_send_idx = nullptr;
    bool any = false;                      // some rank moves a row (the caller's global fact: a collective may only be skipped by ALL)
};

namespace ndcn {

int comm_world(const ndcn_comm *c) { return c ? c->world : 1; }

int comm_allreduce_sum_f64(ndcn_comm *c, double *d_buf, int n, hipStream_t st) {
    if (!c || c->world <= 1) return NDCN_OK;
    int rc = need_rccl(__func__);
    if (rc) return rc;
    NDCN_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat64, kNcclSum, c->comm, st));
    return NDCN_OK;
}

int64_t halo_plan_n_halo(const ndcn_halo_plan *p) { return p ? p->n_halo : 0; }
int64_t halo_plan_n_send(const ndcn_halo_plan *p) { return p ? p->n_send : 0; }
const int32_t *halo_plan_send_idx(const ndcn_halo_plan *p) { return p ? p->d_send_idx : nullptr; }

int halo_exchange_f32(ndcn_halo_plan *p, const float *X, int H, float *d_pack, float *X_halo, hipStream_t st) {
    if (!p->any) return NDCN_OK;
    int rc = need_rccl(__func__);
    if (rc) return rc;
    if (p->n_send > 0) {
        rc = gather_rows_f32(X, p->d_send_idx, p->n_send, H, d_pack, st);
        if (rc) return rc;
    }
    const int world = p->c->world;
    NDCN_NCCL(rccl().GroupStart());
    // a failing Send / Recv must not leave the group open (every later RCCL call of this thread would join it): remember the
    // first failure, close the group, then report
    int bad = kNcclSuccess;
    const char *what = "";
    for (int q = 0; q < world && bad == kNcclSuccess; ++q) {
        if (p->send_counts[q] > 0) {
            bad = rccl().Send(d_pack + p->send_off[q] * H, (size_t)(p->send_counts[q] * H), kNcclFloat32, q, p->c->comm, st);
            what = "ncclSend";
        }
        if (bad == kNcclSuccess && p->recv_counts[q] > 0) {
            bad = rccl().Recv(X_halo + p->recv_off[q] * H, (size_t)(p->recv_counts[q] * H), kNcclFloat32, q, p->c->comm, st);
            what = "ncclRecv";
        }
    }
    const int end = rccl().GroupEnd();
    if (bad != kNcclSuccess) {
        ndcn::set_error("%s: %s failed: %s", __func__, what, rccl().GetErrorString ? rccl().GetErrorString(bad) : "?");
        return NDCN_EHIP;
    }
    NDCN_NCCL(end);
    return NDCN_OK;
}

}  // namespace ndcn

using namespace ndcn;

extern "C" {

int ndcn_comm_unique_id(char h_id[128]) {
    NDCN_CHECK_ARG(h_id, "null id buffer");
    int rc = need_rccl(__func__);
    if (rc) return rc;
    nccl_unique_id id;
    NDCN_NCCL(rccl().GetUniqueId(&id));
    memcpy(h_id, id.internal, 128);
    return NDCN_OK;
}

int ndcn_comm_create(const char h_id[128], int world, int rank, ndcn_comm **out) {
    NDCN_CHECK_ARG(h_id && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    int rc = need_rccl(__func__);
    if (rc) return rc;
    ndcn_comm *c = new (std::nothrow) ndcn_comm();
    if (!c) { set_error("out of host memory"); return NDCN_EINVAL; }
    nccl_unique_id id;
    memcpy(id.internal, h_id, 128);
    int r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != kNcclSuccess) {
        set_error("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
        delete c;
        return NDCN_EHIP;
    }
    c->world = world; c->rank = rank; c->owned = true;
    *out = c;
    return NDCN_OK;
}

int ndcn_comm_adopt(void *nccl_comm, int world, int rank, ndcn_comm **out) {
    NDCN_CHECK_ARG(nccl_comm && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    ndcn_comm *c = new (std::nothrow) ndcn_comm();
    if (!c) { set_error("out of host memory"); return NDCN_EINVAL; }
    c->comm = static_cast<nccl_comm_t>(nccl_comm); c->world = world; c->rank = rank; c->owned = false;
    *out = c;
    return NDCN_OK;
}

int ndcn_comm_destroy(ndcn_comm *c) {
    if (!c) return NDCN_OK;
    if (c->owned && c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
    return NDCN_OK;
}

int ndcn_comm_allreduce_sum_f64(ndcn_comm *c, double *d_buf, int n, void *stream) {
    NDCN_CHECK_ARG(c && d_buf && n > 0, "bad argument");
    return comm_allreduce_sum_f64(c, d_buf, n, static_cast<hipStream_t>(stream));
}

int ndcn_halo_plan_create(ndcn_comm *c, int64_t n_halo, const int64_t *h_send_counts, const int64_t *h_recv_counts,
                          const int32_t *d_send_idx, int any_rank_moves_rows, ndcn_halo_plan **out) {
    NDCN_CHECK_ARG(c && out && h_send_counts && h_recv_counts && n_halo >= 0, "bad argument");
    ndcn_halo_plan *p = new (std::nothrow) ndcn_halo_plan();
    if (!p) { set_error("out of host memory"); return NDCN_EINVAL; }
    p->c = c;
    p->n_halo = n_halo;
    p->send_counts.assign(h_send_counts, h_send_counts + c->world);
    p->recv_counts.assign(h_recv_counts, h_recv_counts + c->world);
    p->send_off.resize(c->world);
    p->recv_off.resize(c->world);
    int64_t so = 0, ro = 0;
    for (int q = 0; q < c->world; ++q) {
        if (p->send_counts[q] < 0 || p->recv_counts[q] < 0) { delete p; set_error("negative row count"); return NDCN_EINVAL; }
        p->send_off[q] = so; so += p->send_counts[q];
        p->recv_off[q] = ro; ro += p->recv_counts[q];
    }
    if (ro != n_halo) { delete p; set_error("receive counts sum to %lld, n_halo is %lld", (long long)ro, (long long)n_halo); return NDCN_EINVAL; }
    if (so > 0 && !d_send_idx) { delete p; set_error("send index list missing"); return NDCN_EINVAL; }
    p->n_send = so;
    p->d_send_idx = d_send_idx;
    p->any = any_rank_moves_rows != 0;
    *out = p;
    return NDCN_OK;
}

int ndcn_halo_plan_destroy(ndcn_halo_plan *p) {
    delete p;
    return NDCN_OK;
}

int ndcn_halo_exchange_f32(ndcn_halo_plan *p, const float *X, int H, float *d_pack, float *X_halo, void *stream) {
    NDCN_CHECK_ARG(p && H > 0, "bad argument");
    NDCN_CHECK_ARG(p->n_send == 0 || (X && d_pack), "pack buffer / panel missing");
    NDCN_CHECK_ARG(p->n_halo == 0 || X_halo, "halo panel missing");
    return halo_exchange_f32(p, X, H, d_pack, X_halo, static_cast<hipStream_t>(stream));
}

}  // extern "C"
