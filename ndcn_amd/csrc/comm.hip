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
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <new>
#include <string>
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

// ---- host-staged loopback transport (TEST infrastructure; ndcn_comm_create_loopback) -----------------------------------------
// RCCL refuses two ranks on one device, and the boxes this is developed on have one: the device-resident sharded solver
// (ndcn_solver_desc::shard) could therefore never run with world > 1 before an 8-GPU node appears.  The loopback communicator has
// the SAME call sequence as the RCCL one - ndcn_halo_exchange_f32 = pack kernel + one send / receive per peer, the 16-byte
// all-reduce - with POSIX shared memory underneath: a message is a shared-memory object of its own (no capacity to size, sends
// never block), announced through a per-pair counter in the communicator's header segment; the all-reduce adds the ranks' slots in
// rank order on every rank (the same bits everywhere).  Everything is synchronous on the caller's stream (stream synchronise,
// device -> host -> shared memory -> host -> device): correctness of the call sequence, not speed.  Waits are bounded (120 s).
constexpr int kLoopMaxWorld = 64, kLoopMaxReduce = 8;
struct LoopHeader {
    std::atomic<unsigned long long> sent[kLoopMaxWorld][kLoopMaxWorld];     // messages src -> dst published so far
    std::atomic<unsigned long long> bar_count, bar_gen;
    double slot[kLoopMaxWorld][kLoopMaxReduce];
};
struct Loopback {
    std::string name;
    LoopHeader *hdr = nullptr;
    unsigned long long recvd[kLoopMaxWorld] = {};
    unsigned long long sent_local[kLoopMaxWorld] = {};
};

static double loop_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
template <class Pred>
static bool loop_wait(Pred done) {
    const double t0 = loop_now();
    while (!done()) {
        if (loop_now() - t0 > 120.0) return false;
        usleep(50);
    }
    return true;
}
static std::string loop_msg_name(const Loopback &l, int src, int dst, unsigned long long seq) {
    return "/" + l.name + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
}
static int loop_barrier(Loopback &l, int world) {
    const unsigned long long g = l.hdr->bar_gen.load(std::memory_order_acquire);
    if (l.hdr->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned long long)world) {
        l.hdr->bar_count.store(0, std::memory_order_relaxed);
        l.hdr->bar_gen.fetch_add(1, std::memory_order_acq_rel);
        return NDCN_OK;
    }
    if (!loop_wait([&] { return l.hdr->bar_gen.load(std::memory_order_acquire) != g; })) {
        ndcn::set_error("loopback communicator: a peer did not reach the barrier within 120 s");
        return NDCN_EHIP;
    }
    return NDCN_OK;
}
static int loop_send(Loopback &l, int me, int dst, const float *d_src, size_t n_floats, hipStream_t st) {
    const size_t bytes = n_floats * sizeof(float);
    const std::string nm = loop_msg_name(l, me, dst, l.sent_local[dst]);
    const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); ndcn::set_error("loopback send: shm_open / ftruncate(%s) failed", nm.c_str()); return NDCN_EHIP; }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { ndcn::set_error("loopback send: mmap failed"); return NDCN_EHIP; }
    const hipError_t e = hipMemcpyAsync(m, d_src, bytes, hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    munmap(m, bytes);
    if (e != hipSuccess || e2 != hipSuccess) { ndcn::set_error("loopback send: device -> host copy failed"); return NDCN_EHIP; }
    ++l.sent_local[dst];
    l.hdr->sent[me][dst].fetch_add(1, std::memory_order_release);
    return NDCN_OK;
}
static int loop_recv(Loopback &l, int me, int src, float *d_dst, size_t n_floats, hipStream_t st) {
    const size_t bytes = n_floats * sizeof(float);
    const unsigned long long seq = l.recvd[src];
    if (!loop_wait([&] { return l.hdr->sent[src][me].load(std::memory_order_acquire) > seq; })) {
        ndcn::set_error("loopback recv: rank %d did not send message %llu within 120 s", src, seq);
        return NDCN_EHIP;
    }
    const std::string nm = loop_msg_name(l, src, me, seq);
    const int fd = shm_open(nm.c_str(), O_RDONLY, 0600);
    if (fd < 0) { ndcn::set_error("loopback recv: shm_open(%s) failed", nm.c_str()); return NDCN_EHIP; }
    void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { ndcn::set_error("loopback recv: mmap failed"); return NDCN_EHIP; }
    const hipError_t e = hipMemcpyAsync(d_dst, m, bytes, hipMemcpyHostToDevice, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    munmap(m, bytes);
    shm_unlink(nm.c_str());
    ++l.recvd[src];
    if (e != hipSuccess || e2 != hipSuccess) { ndcn::set_error("loopback recv: host -> device copy failed"); return NDCN_EHIP; }
    return NDCN_OK;
}

}  // namespace

struct ndcn_comm {
    nccl_comm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
    Loopback *loop = nullptr;              // the host-staged test transport instead of RCCL
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
    if (c->loop) {
        if (n > kLoopMaxReduce) { set_error("loopback all-reduce: at most %d doubles", kLoopMaxReduce); return NDCN_EINVAL; }
        Loopback &l = *c->loop;
        double mine[kLoopMaxReduce], sum[kLoopMaxReduce] = {};
        NDCN_HIP(hipMemcpyAsync(mine, d_buf, n * sizeof(double), hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < n; ++i) l.hdr->slot[c->rank][i] = mine[i];
        int rc = loop_barrier(l, c->world);                  // every slot written
        if (rc) return rc;
        for (int q = 0; q < c->world; ++q)
            for (int i = 0; i < n; ++i) sum[i] += l.hdr->slot[q][i];        // rank order: the same bits on every rank
        if ((rc = loop_barrier(l, c->world))) return rc;     // every slot read before the next reduction overwrites it
        NDCN_HIP(hipMemcpyAsync(d_buf, sum, n * sizeof(double), hipMemcpyHostToDevice, st));
        NDCN_HIP(hipStreamSynchronize(st));
        return NDCN_OK;
    }
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
    int rc = p->c->loop ? NDCN_OK : need_rccl(__func__);
    if (rc) return rc;
    if (p->n_send > 0) {
        rc = gather_rows_f32(X, p->d_send_idx, p->n_send, H, d_pack, st);
        if (rc) return rc;
    }
    const int world = p->c->world;
    if (p->c->loop) {                                        // the same pack + per-peer send / receive, host-staged: all sends, then all receives
        Loopback &l = *p->c->loop;
        for (int q = 0; q < world; ++q)
            if (p->send_counts[q] > 0 && (rc = loop_send(l, p->c->rank, q, d_pack + p->send_off[q] * H, (size_t)(p->send_counts[q] * H), st))) return rc;
        for (int q = 0; q < world; ++q)
            if (p->recv_counts[q] > 0 && (rc = loop_recv(l, p->c->rank, q, X_halo + p->recv_off[q] * H, (size_t)(p->recv_counts[q] * H), st))) return rc;
        return NDCN_OK;
    }
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

int ndcn_comm_create_loopback(const char *name, int world, int rank, ndcn_comm **out) {
    NDCN_CHECK_ARG(name && name[0] && strlen(name) < 100 && out && world >= 1 && world <= kLoopMaxWorld && rank >= 0 && rank < world, "bad argument");
    ndcn_comm *c = new (std::nothrow) ndcn_comm();
    Loopback *l = new (std::nothrow) Loopback();
    if (!c || !l) { delete c; delete l; set_error("out of host memory"); return NDCN_EINVAL; }
    l->name = name;
    const std::string hn = "/" + l->name + "_hdr";
    const int fd = shm_open(hn.c_str(), O_CREAT | O_RDWR, 0600);             // every rank: whoever comes first creates it (zero-filled)
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(LoopHeader)) != 0) {
        if (fd >= 0) close(fd);
        delete c; delete l;
        set_error("loopback communicator: shm_open / ftruncate(%s) failed", hn.c_str());
        return NDCN_EHIP;
    }
    void *m = mmap(nullptr, sizeof(LoopHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; delete l; set_error("loopback communicator: mmap failed"); return NDCN_EHIP; }
    l->hdr = static_cast<LoopHeader *>(m);
    c->world = world; c->rank = rank; c->owned = true; c->loop = l;
    const int rc = loop_barrier(*l, world);                  // everybody has mapped the header before anybody uses (or unlinks) it
    if (rc) { munmap(m, sizeof(LoopHeader)); delete c; delete l; return rc; }
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
    if (c->loop) {
        if (c->loop->hdr) munmap(c->loop->hdr, sizeof(LoopHeader));
        shm_unlink(("/" + c->loop->name + "_hdr").c_str());     // (the first to leave removes the name; the mappings of the others live on)
        delete c->loop;
    } else if (c->owned && c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
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
