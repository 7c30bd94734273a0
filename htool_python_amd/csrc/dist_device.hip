// dist_device.hip -- the exchange step of the row-distributed product, inside the library: RCCL over xGMI.
//
// Replaces the MPI_Allgatherv of htool::add_distributed_operator_vector_product_global_to_global
// (src/htool/distributed_operator/distributed_operator.hpp:33,55; communicator caster src/htool/misc/wrapper_mpi.hpp:28-55).
// One process per GPU; rank p owns the rows of partition p (DefaultApproximationBuilder, utility.hpp:26).  In a
// GPU-resident loop every rank keeps only its slice of a vector, so the exchange is an all-gather of the x slices
// BEFORE the product:
//     ncclAllGather (zero-copy when all slices have the same length, on padded equal slices otherwise)
//  -> one compaction kernel (padded slices -> the contiguous cluster-numbered x; not P copies)
//  -> the local product in cluster numbering (device.hip), all on ONE stream: no host synchronisation in between.
// The message is small (N s / P bytes per rank: 1 MB at N = 10^6, P = 8), so the step is latency-bound; RCCL runs
// it as a direct exchange over the point-to-point xGMI links.
// The communicator is created by the library from a unique id that the host language broadcasts (the NCCL bootstrap
// pattern), or wraps one the caller already has.
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <memory>

#include "capi_internal.hpp"
#include "device_internal.hpp"

using namespace hm;

#define RCCL_OK(call)                                                                                                          \
    do {                                                                                                                       \
        ncclResult_t r_ = (call);                                                                                              \
        if (r_ != ncclSuccess) throw hm::Error(hm::strprintf("RCCL error %s at %s:%d (%s)", ncclGetErrorString(r_), __FILE__, __LINE__, #call)); \
    } while (0)

namespace {

struct RcclComm {
    ncclComm_t comm = nullptr;
    bool owned = true;
    int rank = 0, size = 1, device = 0;
    // staging buffers of the host-buffer exchange (htool_comm.allgatherv on an RCCL communicator)
    void *h_send = nullptr, *h_recv = nullptr;
    size_t h_cap = 0;
    hipStream_t stream = nullptr;
    ~RcclComm() {
        if (h_send) (void)hipFree(h_send);
        if (h_recv) (void)hipFree(h_recv);
        if (stream) (void)hipStreamDestroy(stream);
        if (comm && owned) (void)ncclCommDestroy(comm);
    }
};

// host-buffer all-gather on an RCCL communicator: staged through device buffers (used by the replicated-vector API)
int rccl_host_allgatherv(void *ctx, const void *send, int64_t send_bytes, void *recv, const int64_t *recv_bytes, const int64_t *displs) {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    try {
        HIP_OK(hipSetDevice(c->device));
        int64_t pad = 0;
        for (int p = 0; p < c->size; p++) pad = std::max(pad, recv_bytes[p]);
        pad = (pad + 15) / 16 * 16;
        if ((size_t)pad > c->h_cap) {
            if (c->h_send) (void)hipFree(c->h_send);
            if (c->h_recv) (void)hipFree(c->h_recv);
            c->h_send = c->h_recv = nullptr;
            HIP_OK(hipMalloc(&c->h_send, (size_t)pad));
            HIP_OK(hipMalloc(&c->h_recv, (size_t)pad * c->size));
            c->h_cap = (size_t)pad;
        }
        if (!c->stream) HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_OK(hipMemcpyAsync(c->h_send, send, (size_t)send_bytes, hipMemcpyHostToDevice, c->stream));
        RCCL_OK(ncclAllGather(c->h_send, c->h_recv, (size_t)pad, ncclUint8, c->comm, c->stream));
        for (int p = 0; p < c->size; p++)
            HIP_OK(hipMemcpyAsync((char *)recv + displs[p], (const char *)c->h_recv + (size_t)p * pad, (size_t)recv_bytes[p], hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    } catch (const std::exception &e) {
        htool_error_slot() = e.what();
        return 1;
    }
    return 0;
}

// htool_comm.allgather_device of an RCCL communicator: equal slices on device buffers, in stream order
int rccl_allgather_device(void *ctx, const void *send_dev, void *recv_dev, int64_t bytes, void *stream) {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    try {
        HM_CHECK(c && c->comm, "RCCL communicator already destroyed");
        RCCL_OK(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, (hipStream_t)stream));
    } catch (const std::exception &e) {
        htool_error_slot() = e.what();
        return 1;
    }
    return 0;
}

// htool_comm.reduce_scatter_device of an RCCL communicator: sums of doubles, chunk `rank` of everybody's buffer, in stream order
int rccl_reduce_scatter_device(void *ctx, const void *send_dev, void *recv_dev, int64_t count, void *stream) {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    try {
        HM_CHECK(c && c->comm, "RCCL communicator already destroyed");
        RCCL_OK(ncclReduceScatter(send_dev, recv_dev, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)stream));
    } catch (const std::exception &e) {
        htool_error_slot() = e.what();
        return 1;
    }
    return 0;
}

// the inverse of compact_slices_kernel: z_full[c][displs[p] : +counts[p])  ->  padded[p][c][0 : pad), zeros behind the slice
template <typename T>
__global__ void expand_slices_kernel(const T *__restrict__ z_full, T *__restrict__ padded, const int *__restrict__ counts, const int *__restrict__ displs, int pad, int mu,
                                     long long ldz) {
    const int p = blockIdx.y, c = blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pad) return;
    T v;
    if (i < counts[p]) v = z_full[(long long)c * ldz + displs[p] + i];
    else { double *q = reinterpret_cast<double *>(&v); for (size_t k = 0; k < sizeof(T) / 8; k++) q[k] = 0.0; }
    padded[((long long)p * mu + c) * pad + i] = v;
}

// gathered[p][c][0 : pad)  ->  x_full[c][displs[p] : displs[p] + counts[p])   (one launch for all ranks and columns)
template <typename T>
__global__ void compact_slices_kernel(const T *__restrict__ gathered, T *__restrict__ x_full, const int *__restrict__ counts, const int *__restrict__ displs, int pad, int mu,
                                      long long ldx) {
    const int p = blockIdx.y, c = blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < counts[p]) x_full[(long long)c * ldx + displs[p] + i] = gathered[((long long)p * mu + c) * pad + i];
}

} // namespace

struct DistDeviceState {
    int device = 0;
    void *send = nullptr, *recv = nullptr, *x_full = nullptr;
    int *counts = nullptr, *displs = nullptr;
    int pad = 0, mu_cap = 0;
    bool equal = false;
    // pinned host buffers of the host-staged exchange (communicators without allgather_device)
    void *h_send = nullptr, *h_recv = nullptr;
    size_t h_cap = 0;
    ~DistDeviceState() {
        (void)hipSetDevice(device);
        for (void *p : {send, recv, x_full, (void *)counts, (void *)displs}) if (p) (void)hipFree(p);
        if (h_send) (void)hipHostFree(h_send);
        if (h_recv) (void)hipHostFree(h_recv);
    }
};
void dist_device_free(DistDeviceState *s) { delete s; }

static bool force_padded() { // tests: take the padded-slices path even for equal slices / one rank
    const char *f = getenv("HTOOL_DIST_FORCE_PADDED");
    return f && f[0] == '1';
}

static DistDeviceState *dist_state(htool_distributed *d, int mu) {
    const HMatrix &H = d->hmat->H;
    const size_t es = H.is_complex ? 16 : 8;
    const int P = d->comm.size;
    if (d->dev && d->dev->mu_cap >= mu) return d->dev;
    HM_CHECK((int)d->s_counts.size() == P, "distributed device product: the source cluster tree carries no partition of the communicator's size");
    std::unique_ptr<DistDeviceState> s(new DistDeviceState);
    HIP_OK(hipGetDevice(&s->device));
    std::vector<int> cnt(P), dsp(P);
    int pad = 0;
    bool equal = true;
    for (int p = 0; p < P; p++) {
        cnt[p] = (int)d->s_counts[p];
        dsp[p] = (int)d->s_displs[p];
        pad = std::max(pad, cnt[p]);
        // zero-copy needs rank p's slice at p * pad of the gathered vector: equal counts AND partitions in rank order
        equal = equal && cnt[p] == cnt[0] && d->s_displs[p] == (int64_t)p * cnt[0];
    }
    if (force_padded()) equal = false;
    s->pad = pad;
    s->equal = equal;
    s->mu_cap = mu;
    const size_t ns = (size_t)d->sc->n_points;
    HIP_OK(hipMalloc(&s->send, std::max<size_t>((size_t)pad * mu, 1) * es));
    HIP_OK(hipMalloc(&s->recv, std::max<size_t>((size_t)pad * mu * P, 1) * es));
    HIP_OK(hipMalloc(&s->x_full, std::max<size_t>(ns * mu, 1) * es));
    HIP_OK(hipMalloc((void **)&s->counts, sizeof(int) * P));
    HIP_OK(hipMalloc((void **)&s->displs, sizeof(int) * P));
    HIP_OK(hipMemcpy(s->counts, cnt.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(s->displs, dsp.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    dist_device_free(d->dev);
    d->dev = s.release();
    return d->dev;
}

// the exchange step: every rank contributes `bytes` bytes at send_dev, recv_dev receives rank p's at p * bytes (stream order)
static void gather_equal_slices(htool_distributed *d, DistDeviceState *s, const void *send_dev, void *recv_dev, size_t bytes, hipStream_t st) {
    const int P = d->comm.size;
    if (d->comm.allgather_device) { // RCCL over xGMI (or the host language's own device all-gather)
        const int rc = d->comm.allgather_device(d->comm.ctx, send_dev, recv_dev, (int64_t)bytes, (void *)st);
        HM_CHECK(rc == 0, std::string("distributed device product: allgather_device failed: ") + htool_error_slot());
        return;
    }
    // host-staged (several ranks on one GPU): device -> pinned host, the communicator's host all-gather, pinned host -> device
    HM_CHECK(d->comm.allgatherv != nullptr, "distributed device product: the communicator has neither allgather_device (htool_comm_init_rccl / htool_comm_wrap_rccl) nor allgatherv");
    if (bytes > s->h_cap) {
        if (s->h_send) (void)hipHostFree(s->h_send);
        if (s->h_recv) (void)hipHostFree(s->h_recv);
        s->h_send = s->h_recv = nullptr;
        s->h_cap = 0;
        HIP_OK(hipHostMalloc(&s->h_send, bytes, hipHostMallocDefault));
        HIP_OK(hipHostMalloc(&s->h_recv, bytes * P, hipHostMallocDefault));
        s->h_cap = bytes;
    }
    HIP_OK(hipMemcpyAsync(s->h_send, send_dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st)); // (also: the host -> device copy of the previous call has left h_recv)
    std::vector<int64_t> cnt((size_t)P, (int64_t)bytes), dsp((size_t)P);
    for (int p = 0; p < P; p++) dsp[p] = (int64_t)p * (int64_t)bytes;
    const int rc = d->comm.allgatherv(d->comm.ctx, s->h_send, (int64_t)bytes, s->h_recv, cnt.data(), dsp.data());
    HM_CHECK(rc == 0, "distributed device product: allgatherv failed");
    HIP_OK(hipMemcpyAsync(recv_dev, s->h_recv, bytes * P, hipMemcpyHostToDevice, st));
}

template <typename T>
static void launch_compact(const void *gathered, void *x_full, const int *counts_dev, const int *displs_dev, int P, int pad, int mu, long long ldx, hipStream_t st) {
    if (pad <= 0 || P <= 0 || mu <= 0) return;
    HM_CHECK(P <= 65535 && mu <= 65535, "compaction: too many ranks / columns for one launch");
    const dim3 grid((unsigned)((pad + 255) / 256), (unsigned)P, (unsigned)mu), block(256);
    hipLaunchKernelGGL(compact_slices_kernel<T>, grid, block, 0, st, (const T *)gathered, (T *)x_full, counts_dev, displs_dev, pad, mu, ldx);
    HIP_OK(hipGetLastError());
}

static bool single_owner(const htool_distributed *d) { // one rank owns everything and nothing asks for an exchange
    return d->comm.size == 1 && !d->comm.allgather_device && !force_padded();
}

// Y_local = A[rows of this rank, :] X with X given by its local slices: column c of X_local holds this rank's part
// (source partition `rank`, cluster numbering) at X_local + c * ldx; Y_local + c * ldy receives the rank's rows.
static void dist_matmat_device(htool_distributed *d, const void *X_local, int64_t ldx, void *Y_local, int64_t ldy, int mu, hipStream_t caller_stream) {
    hipStream_t st = caller_stream;
    const HMatrix &H = d->hmat->H;
    const size_t es = H.is_complex ? 16 : 8;
    const int P = d->comm.size, rank = d->comm.rank;
    HM_CHECK(mu >= 1, "mu must be >= 1");
    HM_CHECK(H.dev != nullptr, "H-matrix has no device data");
    if (!st) st = H.dev->stream; // the exchange and the product have to share one stream
    if (single_owner(d)) { // the slice is the whole vector
        device_matmat_device(H, X_local, (long long)ldx, Y_local, (long long)ldy, mu, 1, caller_stream);
        return;
    }
    DistDeviceState *s = dist_state(d, mu);
    HIP_OK(hipSetDevice(s->device));
    HM_CHECK(rank >= 0 && rank < P, "distributed device product: rank out of range");
    const int mine = (int)d->s_counts[rank];
    const size_t ns = (size_t)d->sc->n_points;
    if (s->equal && mu == 1) {
        // equal slices: gather straight from the caller's buffer into the contiguous vector (displs[p] = p * pad)
        gather_equal_slices(d, s, X_local, s->x_full, (size_t)s->pad * es, st);
    } else {
        if (mu == 1) HIP_OK(hipMemcpyAsync(s->send, X_local, (size_t)mine * es, hipMemcpyDeviceToDevice, st));
        else {
            HM_CHECK(ldx >= mine, "distributed device product: ldx is smaller than this rank's slice");
            HIP_OK(hipMemcpy2DAsync(s->send, (size_t)s->pad * es, X_local, (size_t)ldx * es, (size_t)mine * es, (size_t)mu, hipMemcpyDeviceToDevice, st));
        }
        gather_equal_slices(d, s, s->send, s->recv, (size_t)s->pad * mu * es, st);
        if (H.is_complex) launch_compact<double2>(s->recv, s->x_full, s->counts, s->displs, P, s->pad, mu, (long long)ns, st);
        else launch_compact<double>(s->recv, s->x_full, s->counts, s->displs, P, s->pad, mu, (long long)ns, st);
    }
    device_matmat_device(H, s->x_full, (long long)ns, Y_local, (long long)ldy, mu, 1, caller_stream); // (NULL: the operator's own stream, = st)
}

// sum over the ranks of everybody's chunk `rank`: send_dev holds P chunks of `count` doubles, recv_dev receives one (stream order)
static void reduce_scatter_doubles(htool_distributed *d, DistDeviceState *s, const void *send_dev, void *recv_dev, size_t count, hipStream_t st) {
    const int P = d->comm.size, rank = d->comm.rank;
    if (d->comm.reduce_scatter_device) {
        const int rc = d->comm.reduce_scatter_device(d->comm.ctx, send_dev, recv_dev, (int64_t)count, (void *)st);
        HM_CHECK(rc == 0, std::string("distributed device product: reduce_scatter_device failed: ") + htool_error_slot());
        return;
    }
    // host-staged (several ranks on one GPU): everybody's whole buffer is gathered on the host, the own chunk summed in rank order
    HM_CHECK(d->comm.allgatherv != nullptr, "distributed device product: the communicator has neither reduce_scatter_device (htool_comm_init_rccl / htool_comm_wrap_rccl) nor allgatherv");
    const size_t bytes = count * P * sizeof(double);
    if (bytes > s->h_cap) {
        if (s->h_send) (void)hipHostFree(s->h_send);
        if (s->h_recv) (void)hipHostFree(s->h_recv);
        s->h_send = s->h_recv = nullptr;
        s->h_cap = 0;
        HIP_OK(hipHostMalloc(&s->h_send, bytes, hipHostMallocDefault));
        HIP_OK(hipHostMalloc(&s->h_recv, bytes * P, hipHostMallocDefault));
        s->h_cap = bytes;
    }
    HIP_OK(hipMemcpyAsync(s->h_send, send_dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<int64_t> cnt((size_t)P, (int64_t)bytes), dsp((size_t)P);
    for (int p = 0; p < P; p++) dsp[p] = (int64_t)p * (int64_t)bytes;
    const int rc = d->comm.allgatherv(d->comm.ctx, s->h_send, (int64_t)bytes, s->h_recv, cnt.data(), dsp.data());
    HM_CHECK(rc == 0, "distributed device product: allgatherv failed");
    double *acc = (double *)s->h_send; // (reused: the own chunk, summed over the ranks in rank order)
    const double *all = (const double *)s->h_recv;
    for (size_t i = 0; i < count; i++) {
        double v = 0;
        for (int p = 0; p < P; p++) v += all[(size_t)p * count * P + (size_t)rank * count + i];
        acc[i] = v;
    }
    HIP_OK(hipMemcpyAsync(recv_dev, acc, count * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st)); // (acc is overwritten by the next call's device -> host copy)
}

// Y_local = op(A) X restricted to this rank's source slice, X given by the rows this rank owns (see the header)
static void dist_matmat_device_trans(htool_distributed *d, char trans, const void *X_local, int64_t ldx, void *Y_local, int64_t ldy, int mu, hipStream_t caller_stream) {
    hipStream_t st = caller_stream;
    const HMatrix &H = d->hmat->H;
    const size_t es = H.is_complex ? 16 : 8;
    const int P = d->comm.size, rank = d->comm.rank;
    HM_CHECK(mu >= 1, "mu must be >= 1");
    HM_CHECK(trans == 'T' || trans == 'C', "distributed transposed product: trans must be 'T' or 'C'");
    HM_CHECK(H.dev != nullptr, "H-matrix has no device data");
    if (!st) st = H.dev->stream;
    if (d->comm.size == 1 && !d->comm.reduce_scatter_device && !force_padded()) { // one rank owns everything
        device_matmat_device(H, X_local, (long long)ldx, Y_local, (long long)ldy, mu, 1, caller_stream, trans);
        return;
    }
    DistDeviceState *s = dist_state(d, mu);
    HIP_OK(hipSetDevice(s->device));
    const int mine = (int)d->s_counts[rank];
    const size_t ns = (size_t)d->sc->n_points;
    const size_t dbl = es / sizeof(double);
    // the local block applied transposed: a full-length vector (cluster numbering of the source tree), reusing the buffer of the gathered x
    device_matmat_device(H, X_local, (long long)ldx, s->x_full, (long long)ns, mu, 1, caller_stream, trans);
    if (s->equal && mu == 1) { // rank p's slice sits at p * pad: reduce-scatter straight from the vector into the caller's buffer
        reduce_scatter_doubles(d, s, s->x_full, Y_local, (size_t)s->pad * dbl, st);
        return;
    }
    const dim3 grid((unsigned)((s->pad + 255) / 256), (unsigned)P, (unsigned)mu), block(256);
    if (s->pad > 0) {
        if (H.is_complex) hipLaunchKernelGGL(expand_slices_kernel<double2>, grid, block, 0, st, (const double2 *)s->x_full, (double2 *)s->recv, s->counts, s->displs, s->pad, mu, (long long)ns);
        else hipLaunchKernelGGL(expand_slices_kernel<double>, grid, block, 0, st, (const double *)s->x_full, (double *)s->recv, s->counts, s->displs, s->pad, mu, (long long)ns);
        HIP_OK(hipGetLastError());
    }
    reduce_scatter_doubles(d, s, s->recv, s->send, (size_t)s->pad * mu * dbl, st); // [p][c][pad] summed -> [c][pad]
    if (mine > 0) {
        if (mu == 1) HIP_OK(hipMemcpyAsync(Y_local, s->send, (size_t)mine * es, hipMemcpyDeviceToDevice, st));
        else {
            HM_CHECK(ldy >= mine, "distributed transposed product: ldy is smaller than this rank's slice");
            HIP_OK(hipMemcpy2DAsync(Y_local, (size_t)ldy * es, s->send, (size_t)s->pad * es, (size_t)mine * es, (size_t)mu, hipMemcpyDeviceToDevice, st));
        }
    }
}

extern "C" {

int htool_distributed_matmat_device_trans(htool_distributed *d, char trans, const void *X_local_dev, int64_t ldx, void *Y_local_dev, int64_t ldy, int mu, void *stream) {
    API_BEGIN
    dist_matmat_device_trans(d, trans, X_local_dev, ldx, Y_local_dev, ldy, mu, (hipStream_t)stream);
    API_END
}

int htool_rccl_get_unique_id(void *id128) {
    API_BEGIN
    static_assert(sizeof(ncclUniqueId) == HTOOL_RCCL_UNIQUE_ID_BYTES, "unique id size");
    HM_CHECK(id128 != nullptr, "htool_rccl_get_unique_id: null argument");
    ncclUniqueId id;
    RCCL_OK(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    API_END
}

int htool_comm_init_rccl(const void *id128, int rank, int size, htool_comm *out) {
    API_BEGIN
    HM_CHECK(id128 && out && size >= 1 && rank >= 0 && rank < size, "htool_comm_init_rccl: bad argument");
    std::unique_ptr<RcclComm> c(new RcclComm);
    HIP_OK(hipGetDevice(&c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCL_OK(ncclCommInitRank(&c->comm, size, id, rank));
    c->rank = rank;
    c->size = size;
    out->rank = rank;
    out->size = size;
    out->allgatherv = &rccl_host_allgatherv;
    out->allgather_device = &rccl_allgather_device;
    out->reduce_scatter_device = &rccl_reduce_scatter_device;
    out->rccl = c.get();
    out->ctx = c.release();
    API_END
}

int htool_comm_wrap_rccl(void *nccl_comm, int rank, int size, htool_comm *out) {
    API_BEGIN
    HM_CHECK(nccl_comm && out && size >= 1 && rank >= 0 && rank < size, "htool_comm_wrap_rccl: bad argument");
    std::unique_ptr<RcclComm> c(new RcclComm);
    HIP_OK(hipGetDevice(&c->device));
    c->comm = (ncclComm_t)nccl_comm;
    c->owned = false;
    c->rank = rank;
    c->size = size;
    out->rank = rank;
    out->size = size;
    out->allgatherv = &rccl_host_allgatherv;
    out->allgather_device = &rccl_allgather_device;
    out->reduce_scatter_device = &rccl_reduce_scatter_device;
    out->rccl = c.get();
    out->ctx = c.release();
    API_END
}

void htool_comm_destroy_rccl(htool_comm *c) {
    if (c && c->rccl) {
        delete static_cast<RcclComm *>(c->rccl);
        c->rccl = c->ctx = nullptr;
        c->allgatherv = nullptr;
        c->allgather_device = nullptr;
        c->reduce_scatter_device = nullptr;
    }
}

int htool_distributed_matvec_device(htool_distributed *d, const void *x_local_dev, void *y_local_dev, void *stream) {
    API_BEGIN
    dist_matmat_device(d, x_local_dev, 0, y_local_dev, 0, 1, (hipStream_t)stream);
    API_END
}

int htool_distributed_matmat_device(htool_distributed *d, const void *X_local_dev, int64_t ldx, void *Y_local_dev, int64_t ldy, int mu, void *stream) {
    API_BEGIN
    dist_matmat_device(d, X_local_dev, ldx, Y_local_dev, ldy, mu, (hipStream_t)stream);
    API_END
}

int htool_distributed_exchange_kind(const htool_distributed *d, int mu) {
    if (!d || single_owner(d)) return 0;
    if ((int)d->s_counts.size() != d->comm.size) return -1; // the source tree carries no partition of the communicator's size
    bool equal = !force_padded();
    for (int p = 0; equal && p < d->comm.size; p++) equal = d->s_counts[p] == d->s_counts[0] && d->s_displs[p] == (int64_t)p * d->s_counts[0];
    const int kind = (equal && mu == 1) ? 1 : 2;
    return d->comm.allgather_device ? kind : kind + 2;
}

int htool_debug_compact_slices(const void *gathered_dev, void *x_full_dev, const int *counts, const int *displs, int P, int pad, int mu, int64_t ldx, int is_complex,
                               void *stream) {
    API_BEGIN
    HM_CHECK(gathered_dev && x_full_dev && counts && displs && P >= 1 && pad >= 0 && mu >= 1, "htool_debug_compact_slices: bad argument");
    for (int p = 0; p < P; p++) HM_CHECK(counts[p] >= 0 && counts[p] <= pad && displs[p] >= 0 && (int64_t)displs[p] + counts[p] <= ldx, "htool_debug_compact_slices: slice out of range");
    int *tab = nullptr;
    HIP_OK(hipMalloc((void **)&tab, sizeof(int) * 2 * (size_t)P));
    try {
        HIP_OK(hipMemcpy(tab, counts, sizeof(int) * P, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(tab + P, displs, sizeof(int) * P, hipMemcpyHostToDevice));
        if (is_complex) launch_compact<double2>(gathered_dev, x_full_dev, tab, tab + P, P, pad, mu, (long long)ldx, (hipStream_t)stream);
        else launch_compact<double>(gathered_dev, x_full_dev, tab, tab + P, P, pad, mu, (long long)ldx, (hipStream_t)stream);
        HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    } catch (...) {
        (void)hipFree(tab);
        throw;
    }
    (void)hipFree(tab);
    API_END
}

} // extern "C"

// library warm-up (device.hip: device_warm_up): the first launch of a kernel of this translation unit loads its code object
namespace hm {
__global__ void warm_kernel_dist() {}
void warm_up_dist() { hipLaunchKernelGGL(warm_kernel_dist, dim3(1), dim3(64), 0, 0); }
} // namespace hm
