// The exchange step of the multi-GPU path for a C++ host (one process per GPU): what iv_slam_amd/dist.py + bench.py do through
// torch.distributed, written against the C-ABI (include/ivfront.h), the HIP runtime and RCCL (<rccl/rccl.h>; "nccl" IS RCCL on ROCm).
// Compile-checked by tests/test_adapter_compiles.py (no GPU needed); INTEGRATION.md section 6 shows the part between the markers.
// Reference counterpart: none -- the reference tracks one frame at a time on one CPU thread (ORB/src/Tracking.cc:1303-1330).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstddef>
#include <cstdint>
#include <vector>
#include "ivfront.h"

#define IVX_HIP(e) do { if ((e) != hipSuccess) return IVF_E_NO_DEVICE; } while (0)
#define IVX_NCCL(e) do { if ((e) != ncclSuccess) return IVF_E_NO_DEVICE; } while (0)
#define IVX_IVF(e) do { const int rc_ = (e); if (rc_ != IVF_OK) return rc_; } while (0)

// [integration-snippet-begin]
// Frame k of the stream is extracted on rank k mod G; a batch is P frames per rank = G * P consecutive frames.  After the all-gather
// record slot r * P + j holds global frame j * G + r; one more slot, index G * P, holds the LAST frame of the previous batch (the carry),
// so that every frame has its predecessor: Tracking::TrackWithMotionModel runs for every frame (Tracking.cc:1303-1330).
struct IvfExchange {
    static const int kDepth = 3;                     // batches in flight = the front end's three contexts
    int world = 1, rank = 0, P = 0, N = 0;           // ranks, this rank, pairs per rank and batch, nfeatures
    size_t rec = 0;                                  // ivf_track_record_bytes(N)
    ncclComm_t comm = nullptr;
    uint8_t* send[kDepth] = {};                      // [P records]: this rank's block of batch k % kDepth
    uint8_t* records[kDepth] = {};                   // [G * P + 1 records]: gathered block + carry slot
    int32_t* d_pairs = nullptr;                      // [P][2] (last, cur) slots of the frames THIS rank extracted
    int n_pairs = 0;
    hipEvent_t published[kDepth] = {}, released[kDepth] = {};
    bool havePublished[kDepth] = {}, haveReleased[kDepth] = {};
    long long batch = 0;
};

// pair table of dist.track_pairs(world, rank, P, carry=True): the frame before slot (r, j) is (r - 1, j), or (G - 1, j - 1) for r = 0,
// or the carry slot for the batch's very first frame
static std::vector<int32_t> ivx_track_pairs(int G, int rank, int P)
{
    std::vector<int32_t> t;
    for (int j = 0; j < P; j++) {
        if (rank > 0) { t.push_back((rank - 1) * P + j); t.push_back(rank * P + j); }
        else if (j > 0) { t.push_back((G - 1) * P + j - 1); t.push_back(j); }
        else { t.push_back(G * P); t.push_back(0); }
    }
    return t;
}

int ivx_exchange_create(IvfExchange* x, ncclComm_t comm, int world, int rank, int pairs_per_rank, int nfeatures)
{
    x->comm = comm; x->world = world; x->rank = rank; x->P = pairs_per_rank; x->N = nfeatures;
    x->rec = ivf_track_record_bytes(nfeatures);
    for (int k = 0; k < IvfExchange::kDepth; k++) {
        IVX_HIP(hipMalloc((void**)&x->send[k], x->rec * x->P));
        IVX_HIP(hipMalloc((void**)&x->records[k], x->rec * ((size_t)world * x->P + 1)));
        IVX_HIP(hipMemset(x->records[k], 0, x->rec * ((size_t)world * x->P + 1)));      // an empty carry record: n = 0, no matches
        IVX_HIP(hipEventCreateWithFlags(&x->published[k], hipEventDisableTiming));
        IVX_HIP(hipEventCreateWithFlags(&x->released[k], hipEventDisableTiming));
    }
    const std::vector<int32_t> t = ivx_track_pairs(world, rank, x->P);
    x->n_pairs = (int)t.size() / 2;
    IVX_HIP(hipMalloc((void**)&x->d_pairs, t.size() * sizeof(int32_t)));
    IVX_HIP(hipMemcpy(x->d_pairs, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return IVF_OK;
}

// One batch, after ivf_frontend_run(fe, ...) enqueued its extraction: everything below goes on the INTERNAL stream that batch runs on
// (ivf_frontend_batch_stream(fe, 0)), in order behind it -- no cross-stream wait except the one the boundary pair needs.
//   d_poses [n_pairs][24] (Tcw_last | Tcw_cur_prior, row-major 3x4) or NULL; d_assign [n_pairs][N], d_nmatches [n_pairs]: results.
int ivx_exchange_step(IvfExchange* x, ivf_frontend* fe, ivf_tracker* trk, const float* d_poses, int32_t* d_assign, int32_t* d_nmatches)
{
    const int k = (int)(x->batch % IvfExchange::kDepth), nxt = (int)((x->batch + 1) % IvfExchange::kDepth);
    hipStream_t st = (hipStream_t)ivf_frontend_batch_stream(fe, 0);
    const size_t gathered = (size_t)x->world * x->P;
    size_t rec = 0;
    // 1. this rank's P records {n, kps, desc, uRight, depth} behind the extraction, on its own stream
    IVX_IVF(ivf_frontend_pack_gather_block(fe, x->world > 1 ? x->send[k] : x->records[k], x->rec * x->P, &rec, IVF_STREAM_OF_BATCH));
    // 2. the exchange: ONE all-gather of fixed-size blocks per batch (rank r's block lands at slot r * P) -- RCCL over xGMI
    if (x->world > 1) IVX_NCCL(ncclAllGather(x->send[k], x->records[k], x->rec * x->P, ncclUint8, x->comm, st));
    // 3. hand the batch's last global frame (slot G * P - 1: rank G - 1's last) to the next batch's carry slot.  That slot was last read by
    //    the tracker step of batch + 1 - kDepth: complete long ago in steady state, so the host query almost never inserts a wait
    if (x->haveReleased[nxt] && hipEventQuery(x->released[nxt]) != hipSuccess) IVX_HIP(hipStreamWaitEvent(st, x->released[nxt], 0));
    IVX_HIP(hipMemcpyAsync(x->records[nxt] + gathered * x->rec, x->records[k] + (gathered - 1) * x->rec, x->rec, hipMemcpyDeviceToDevice, st));
    IVX_HIP(hipEventRecord(x->published[nxt], st)); x->havePublished[nxt] = true;
    // 4. the consumer: Tracking::TrackWithMotionModel's matcher part for this rank's P frames, each against its predecessor wherever that was
    //    extracted; the batch's first frame needs the carry record the PREVIOUS batch published on ITS stream: the one cross-stream wait
    if (x->havePublished[k]) IVX_HIP(hipStreamWaitEvent(st, x->published[k], 0));
    IVX_IVF(ivf_tracker_run(trk, x->records[k], x->rec, (int)gathered + 1, x->d_pairs, x->n_pairs, d_poses, nullptr, nullptr, nullptr,
                            d_assign, d_nmatches, st));
    IVX_HIP(hipEventRecord(x->released[k], st)); x->haveReleased[k] = true;
    x->batch++;
    return IVF_OK;
}
// [integration-snippet-end]

void ivx_exchange_destroy(IvfExchange* x)
{
    for (int k = 0; k < IvfExchange::kDepth; k++) {
        if (x->send[k]) (void)hipFree(x->send[k]);
        if (x->records[k]) (void)hipFree(x->records[k]);
        if (x->published[k]) (void)hipEventDestroy(x->published[k]);
        if (x->released[k]) (void)hipEventDestroy(x->released[k]);
    }
    if (x->d_pairs) (void)hipFree(x->d_pairs);
    *x = IvfExchange();
}

// how a rank gets its communicator: rank 0 makes the id, the launcher (MPI, a file, a socket) hands its 128 bytes to the others
int ivx_comm_create(int world, int rank, const ncclUniqueId* id_from_rank0, ncclComm_t* out)
{
    IVX_NCCL(ncclCommInitRank(out, world, *id_from_rank0, rank));
    return IVF_OK;
}

// ---- C shim for tests/test_gpu_track.py (ctypes): one rank runs the very same step (world = 1: no communicator, no collective) ----
extern "C" {
void* ivx_c_create(int world, int rank, int pairs_per_rank, int nfeatures, void* comm)
{
    IvfExchange* x = new IvfExchange();
    if (ivx_exchange_create(x, (ncclComm_t)comm, world, rank, pairs_per_rank, nfeatures) != IVF_OK) { ivx_exchange_destroy(x); delete x; return nullptr; }
    return x;
}
int ivx_c_step(void* x, void* fe, void* trk, const float* d_poses, int32_t* d_assign, int32_t* d_nmatches)
{
    return ivx_exchange_step((IvfExchange*)x, (ivf_frontend*)fe, (ivf_tracker*)trk, d_poses, d_assign, d_nmatches);
}
const uint8_t* ivx_c_records(void* x, int k) { return ((IvfExchange*)x)->records[k % IvfExchange::kDepth]; }
int ivx_c_pairs(void* x, int32_t* host_pairs, int cap)
{
    IvfExchange* e = (IvfExchange*)x;
    if (cap < 2 * e->n_pairs || hipMemcpy(host_pairs, e->d_pairs, 2 * e->n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return e->n_pairs;
}
void ivx_c_destroy(void* x) { if (x) { ivx_exchange_destroy((IvfExchange*)x); delete (IvfExchange*)x; } }
}
