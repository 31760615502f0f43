// ivf_track.hip -- batched, device-resident tracker step: the consumer of the descriptor all-gather.
//
// For every (last, cur) pair of gather records -- what ivf_frontend_pack_gather_block writes and the RCCL all-gather
// delivers -- one launch sequence runs the matcher part of Tracking::TrackWithMotionModel (ORB/src/Tracking.cc:1303-1330):
//   * UpdateLastFrame's "visual odometry" points (Tracking.cc:1256-1300): the last frame's stereo points, optionally only the
//     close ones + the 100 closest, un-projected with Frame::UnprojectStereo (ORB/src/Frame.cc:958-972);
//   * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono = false) (ORB/src/ORBmatcher.cc:1372-1518): projection
//     into the current frame (:1399-1427), Frame::GetFeaturesInArea windows on the 64x48 grid (ORB/src/Frame.cc:615-668) with the
//     forward / backward / +-1 octave range (:1429-1434), the ORDER-DEPENDENT greedy assignment in last-keypoint order with the
//     stereo-consistency check (:1444-1472), the 30-bin rotation histogram and ComputeThreeMaxima (:1475-1511, :1654-1695);
//   * the retry with the wider window when fewer than 20 matches were found (Tracking.cc:1320-1330).
// Nothing crosses PCIe: records, poses, the per-pair grids, candidate lists and results stay in HBM.
//
// Kernels (all hand-written for gfx950; wave = 64):
//   k_track_prepare  one workgroup per frame pair: AssignFeaturesToGrid of the current record (stable counting sort in LDS),
//                    pose algebra in double (cv::gemm semantics), depth ranking, projection -> query table
//   k_track_window   one wave per (frame pair, last keypoint): window + octave / box filters + Hamming distances,
//                    ballot-compacted, ORDERED candidate list of <= 64 packed entries (index | distance << 16)
//   k_track_greedy   one wave per frame pair walks the queries in order: assignment state, uRight and angles of the current
//                    frame live in LDS, the lists of the next 16 queries are in registers before they are needed, the first
//                    minimum is a DPP row_shr / row_bcast reduction of (distance << 6 | list position); histogram in one VGPR
//                    (lane = bin).  Queries whose window overflowed the list are re-walked in place against the live state.
#include "ivf_device.h"
#include <climits>
#include <cstring>
#include <vector>

using namespace ivf;

namespace {

#define fail ivf::set_error
#define HIPCHK(expr)                                                                                   \
    do { hipError_t e_ = (expr);                                                                        \
         if (e_ != hipSuccess) return fail(IVF_E_NO_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define DEVINL __device__ __forceinline__

constexpr int kGC = 64, kGR = 48;                 // FRAME_GRID_COLS / ROWS (ORB/include/Frame.h:43-44)
constexpr int kListCap = 64;                      // candidates listed per query (one per lane of the greedy wave)
constexpr int kPrefetch = 16;                     // queries whose lists are in registers ahead of the greedy walk
constexpr int kMaxTrackFeatures = 4096;           // LDS state of k_track_greedy: 16 B per keypoint

struct TrackParams {                               // uniform kernel arguments
    int nf, nlevels;
    float scale[kMaxLevels];
    float fx, fy, cx, cy, invfx, invfy, bf, b;
    float minX, minY, maxX, maxY, invW, invH;     // image bounds and mfGridElementWidthInv / HeightInv (Frame.cc:208-209)
    float thDepth;                                // > 0: UpdateLastFrame's close-point rule
    int checkOri, defaultBlocks;
    size_t recBytes;
};

// gather record: {int32 n; int32 pad[3]; ivf_keypoint kps[nf]; uint8 desc[nf][32]; float uright[nf]; float depth[nf]}
DEVINL int rec_count(const uint8_t* r, int nf) { const int n = *(const int*)r; return n < 0 ? 0 : (n > nf ? nf : n); }
DEVINL const ivf_keypoint* rec_kps(const uint8_t* r) { return (const ivf_keypoint*)(r + 16); }
DEVINL const uint8_t* rec_desc(const uint8_t* r, int nf) { return r + 16 + (size_t)nf * 24; }
DEVINL const float* rec_uright(const uint8_t* r, int nf) { return (const float*)(r + 16 + (size_t)nf * 56); }
DEVINL const float* rec_depth(const uint8_t* r, int nf) { return (const float*)(r + 16 + (size_t)nf * 60); }

// per-query record written by k_track_prepare: projection (u, v), ur = u - bf * invzc, and the packed
// {octave, minLevel + 1, maxLevel + 1, flags: bit 0 valid, bit 1 blocks}
struct __attribute__((aligned(16))) Query { float u, v, ur; unsigned bits; };
DEVINL unsigned pack_bits(int oct, int lo, int hi, int valid, int blocks)
{ return (unsigned)oct | ((unsigned)(lo + 1) << 8) | ((unsigned)(hi + 1) << 16) | ((unsigned)valid << 24) | ((unsigned)blocks << 25); }

// cv::gemm on CV_32F operands: double accumulation of (double)a * (double)b, + (double)c, one narrowing
DEVINL void mul_add(const float* R, const float* p, const float* t, float* out)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
        out[i] = (float)((double)R[3 * i] * (double)p[0] + (double)R[3 * i + 1] * (double)p[1] + (double)R[3 * i + 2] * (double)p[2] + (double)t[i]);
}
// -R.t() * t (Frame::UpdatePoseMatrices, Frame.cc:549-555; ORBmatcher.cc:1385)
DEVINL void neg_rt_mul(const float* R, const float* t, float* out)
{
#pragma unroll
    for (int j = 0; j < 3; j++)
        out[j] = (float)(-((double)R[j] * (double)t[0] + (double)R[3 + j] * (double)t[1] + (double)R[6 + j] * (double)t[2]));
}

DEVINL int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// minimum over the wave without a trip through the LDS crossbar: row_shr 1/2/4/8 folds each row of 16 lanes into its lane 15,
// row_bcast15 / row_bcast31 fold the four rows into lane 63
DEVINL unsigned wave_min_u32_dpp(unsigned v)
{
    const int idn = -1;                           // 0xFFFFFFFF: identity of the unsigned minimum
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:1
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:2
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:4
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:8
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;   // row_bcast:15 -> rows 1, 3
    t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Frame::GetFeaturesInArea (Frame.cc:615-668) for one query, walked by one wave 64 candidates at a time in the reference's
// order (grid column ix outer, row iy inner, insertion order inside a bucket -- the buckets of a column are one contiguous run
// of `idx`).  f(ok, i2, dist) is called by all lanes for every step; ok = the lane holds a candidate that passed the octave
// and box filters, dist = its Hamming distance to the query descriptor.
template <class F>
DEVINL void walk_window(const TrackParams& P, const ivf_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                        const int* __restrict__ start, const unsigned short* __restrict__ idx, float x, float y, float r,
                        int minL, int maxL, const uint4 qa, const uint4 qb, int lane, F&& f)
{
    const int x0 = max(0, (int)floorf((x - P.minX - r) * P.invW)), x1 = min(kGC - 1, (int)ceilf((x - P.minX + r) * P.invW));
    const int y0 = max(0, (int)floorf((y - P.minY - r) * P.invH)), y1 = min(kGR - 1, (int)ceilf((y - P.minY + r) * P.invH));
    if (!(x0 < kGC && x1 >= 0 && y0 < kGR && y1 >= 0)) return;
    const bool chk = (minL > 0) || (maxL >= 0);
    for (int ix = x0; ix <= x1; ix++) {
        const int s = start[ix * kGR + y0], e = start[ix * kGR + y1 + 1];
        for (int j0 = s; j0 < e; j0 += 64) {
            const int j = j0 + lane;
            bool ok = j < e;
            int i2 = 0, d = 0;
            if (ok) {
                i2 = idx[j];
                const ivf_keypoint kp = kps[i2];
                if (chk) { if (kp.octave < minL) ok = false; if (maxL >= 0 && kp.octave > maxL) ok = false; }
                if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) ok = false;
                if (ok) {
                    const uint4* cd = (const uint4*)(desc + (size_t)i2 * 32);
                    d = hamming256(cd[0], cd[1], qa, qb);
                }
            }
            f(ok, i2, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_track_prepare: one workgroup per frame pair
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_prepare(TrackParams P, const uint8_t* __restrict__ records, const int2* __restrict__ pairs,
                                                      const float* __restrict__ poses, const uint8_t* __restrict__ pointFlags,
                                                      int* __restrict__ gStart, unsigned short* __restrict__ gIdx,
                                                      Query* __restrict__ queries, int* __restrict__ retryFlag)
{
    __shared__ int cnt[kGC * kGR];
    __shared__ int part[256];
    __shared__ int blk[256];
    __shared__ float s_z[kMaxTrackFeatures];
    __shared__ int s_nStereo, s_nClose;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int2 pr = pairs[p];
    const uint8_t* recL = records + (size_t)pr.x * P.recBytes;
    const uint8_t* recC = records + (size_t)pr.y * P.recBytes;
    const int nL = rec_count(recL, P.nf), nC = rec_count(recC, P.nf);
    if (tid == 0) { retryFlag[p] = 0; s_nStereo = 0; s_nClose = 0; }

    // ---- Frame::AssignFeaturesToGrid of the current frame (Frame.cc:415-430): CSR, buckets in ix-major order, keypoints of a
    // bucket in insertion order (rank inside the bucket = same-bucket keypoints in earlier 256-blocks + earlier threads)
    const ivf_keypoint* kc = rec_kps(recC);
    int* start = gStart + (size_t)p * (kGC * kGR + 1);
    unsigned short* idx = gIdx + (size_t)p * P.nf;
    for (int c = tid; c < kGC * kGR; c += 256) cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](int i) {
        const int px = (int)roundf((kc[i].x - P.minX) * P.invW), py = (int)roundf((kc[i].y - P.minY) * P.invH);   // PosInGrid :672-673
        return (px < 0 || px >= kGC || py < 0 || py >= kGR) ? -1 : px * kGR + py;
    };
    for (int i = tid; i < nC; i += 256) { const int c = cell_of(i); if (c >= 0) atomicAdd(&cnt[c], 1); }
    __syncthreads();
    constexpr int PER = kGC * kGR / 256;
    int local[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { local[k] = sum; sum += cnt[tid * PER + k]; }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int base = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { start[tid * PER + k] = base + local[k]; cnt[tid * PER + k] = base + local[k]; }
    if (tid == 255) start[kGC * kGR] = part[255];
    __syncthreads();
    for (int i0 = 0; i0 < nC; i0 += 256) {
        const int i = i0 + tid;
        const int c = i < nC ? cell_of(i) : -1;
        blk[tid] = c;
        __syncthreads();
        if (c >= 0) {
            int before = 0;
            for (int t = 0; t < tid; t++) before += blk[t] == c ? 1 : 0;
            idx[cnt[c] + before] = (unsigned short)i;
        }
        __syncthreads();
        if (c >= 0) atomicAdd(&cnt[c], 1);
        __syncthreads();
    }

    // ---- poses: Tcw of the two frames (row-major 3x4), identity when none are given (zero-motion prior)
    float Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tl[3] = {0, 0, 0}, Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tc[3] = {0, 0, 0};
    if (poses) {
        const float* Tl = poses + (size_t)pr.x * 12; const float* Tc = poses + (size_t)pr.y * 12;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) { Rl[3 * i + j] = Tl[4 * i + j]; Rc[3 * i + j] = Tc[4 * i + j]; }
            tl[i] = Tl[4 * i + 3]; tc[i] = Tc[4 * i + 3];
        }
    }
    float Rwl[9], Owl[3], twc[3], tlc[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rwl[3 * i + j] = Rl[3 * j + i];                    // mRwc = mRcw.t()
    neg_rt_mul(Rl, tl, Owl);                                                          // mOw = -mRcw.t() * mtcw
    neg_rt_mul(Rc, tc, twc);                                                          // ORBmatcher.cc:1385
    mul_add(Rl, twc, tl, tlc);                                                        // tlc = Rlw * twc + tlw (:1390)
    const bool bForward = tlc[2] > P.b, bBackward = -tlc[2] > P.b;                    // :1392-1393 (bMono = false)

    // ---- which last-frame keypoints carry a point
    const float* zL = rec_depth(recL, P.nf);
    const ivf_keypoint* kl = rec_kps(recL);
    const uint8_t* fl = pointFlags ? pointFlags + (size_t)pr.x * P.nf : nullptr;
    const bool rule = !fl && P.thDepth > 0.0f;
    int rStop = INT_MAX;
    if (rule) {
        // UpdateLastFrame (Tracking.cc:1256-1300): points sorted by (depth, index); all close ones, and at least the 100 closest:
        // the walk stops after the first entry with z > mThDepth once more than 100 have been taken
        int ns = 0, nc = 0;
        for (int i = tid; i < nL; i += 256) { const float z = zL[i]; s_z[i] = z; if (z > 0) { ns++; if (!(z > P.thDepth)) nc++; } }
        atomicAdd(&s_nStereo, ns); atomicAdd(&s_nClose, nc);
        __syncthreads();
        rStop = max(s_nClose, 100);
    }
    Query* Q = queries + (size_t)p * P.nf;
    for (int i = tid; i < nL; i += 256) {
        const float z = zL[i];
        bool has = z > 0;
        int blocks = P.defaultBlocks;
        if (fl) { has = has && (fl[i] & 1); blocks = (fl[i] >> 1) & 1; }
        if (has && rule) {
            int rank = 0;
            for (int j = 0; j < nL; j++) { const float zj = s_z[j]; rank += (zj > 0 && (zj < z || (zj == z && j < i))) ? 1 : 0; }
            has = rank <= rStop;
            blocks = 0;                                                               // new "visual odometry" points have no observations
        }
        Query q; q.u = 0; q.v = 0; q.ur = 0; q.bits = 0;
        if (has) {
            const ivf_keypoint kp = kl[i];
            // Frame::UnprojectStereo (Frame.cc:958-972)
            float x3[3], xw[3], xc[3];
            x3[0] = (kp.x - P.cx) * z * P.invfx; x3[1] = (kp.y - P.cy) * z * P.invfy; x3[2] = z;
            mul_add(Rwl, x3, Owl, xw);
            mul_add(Rc, xw, tc, xc);                                                  // ORBmatcher.cc:1405
            const float invzc = (float)(1.0 / (double)xc[2]);                         // :1409
            if (!(invzc < 0)) {
                const float u = P.fx * xc[0] * invzc + P.cx, v = P.fy * xc[1] * invzc + P.cy;
                if (!(u < P.minX || u > P.maxX) && !(v < P.minY || v > P.maxY)) {
                    const int o = kp.octave;
                    int lo, hi;
                    if (bForward) { lo = o; hi = -1; } else if (bBackward) { lo = 0; hi = o; } else { lo = o - 1; hi = o + 1; }   // :1429-1434
                    q.u = u; q.v = v; q.ur = u - P.bf * invzc;                         // :1453
                    q.bits = pack_bits(o, lo, hi, 1, blocks);
                }
            }
        }
        Q[i] = q;
    }
}

// ------------------------------------------------------------------------------------------------
// k_track_window: one wave per (frame pair, last keypoint)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_window(TrackParams P, const uint8_t* __restrict__ records, const int2* __restrict__ pairs,
                                                     const int* __restrict__ gStart, const unsigned short* __restrict__ gIdx,
                                                     const Query* __restrict__ queries, float th, const int* __restrict__ retryFlag,
                                                     int onlyFlagged, int* __restrict__ count, unsigned* __restrict__ lists)
{
    const int p = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (onlyFlagged && !retryFlag[p]) return;
    const int2 pr = pairs[p];
    const uint8_t* recL = records + (size_t)pr.x * P.recBytes;
    const uint8_t* recC = records + (size_t)pr.y * P.recBytes;
    const int nL = rec_count(recL, P.nf);
    if (i >= nL) return;
    const Query q = queries[(size_t)p * P.nf + i];
    int total = 0;
    if ((q.bits >> 24) & 1) {
        const int o = q.bits & 0xff, lo = (int)((q.bits >> 8) & 0xff) - 1, hi = (int)((q.bits >> 16) & 0xff) - 1;
        const float r = th * P.scale[o];                                              // :1427
        const uint4* qd = (const uint4*)(rec_desc(recL, P.nf) + (size_t)i * 32);       // pMP->GetDescriptor(): the point's one observation
        const uint4 qa = qd[0], qb = qd[1];
        unsigned* L = lists + ((size_t)p * P.nf + i) * kListCap;
        walk_window(P, rec_kps(recC), rec_desc(recC, P.nf), gStart + (size_t)p * (kGC * kGR + 1), gIdx + (size_t)p * P.nf,
                    q.u, q.v, r, lo, hi, qa, qb, lane, [&](bool ok, int i2, int d) {
                        const unsigned long long m = __ballot(ok);
                        if (ok) {
                            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
                            if (pos < kListCap) L[pos] = (unsigned)i2 | ((unsigned)d << 16);
                        }
                        total += __popcll(m);
                    });
    }
    if (lane == 0) count[(size_t)p * P.nf + i] = total;
}

// ------------------------------------------------------------------------------------------------
// k_track_greedy: one wave per frame pair
// ------------------------------------------------------------------------------------------------
constexpr int kNoBlock = 0x10000;                 // flag on an assignment made by a point without observations: it does not block (:1447-1449)
__global__ __launch_bounds__(64) void k_track_greedy(TrackParams P, const uint8_t* __restrict__ records, const int2* __restrict__ pairs,
                                                    const int* __restrict__ gStart, const unsigned short* __restrict__ gIdx,
                                                    const Query* __restrict__ queries, float th, int* __restrict__ retryFlag,
                                                    int onlyFlagged, int retryBelow, const int* __restrict__ count,
                                                    const unsigned* __restrict__ lists, int* __restrict__ assignOut,
                                                    int* __restrict__ nmatchesOut)
{
    extern __shared__ int s_mem[];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (onlyFlagged && !retryFlag[p]) return;
    int* s_assign = s_mem;                                    // [nf] CurrentFrame.mvpMapPoints as last-keypoint indices
    float* s_ur = (float*)(s_mem + P.nf);                     // [nf] CurrentFrame.mvuRight
    float* s_ang = (float*)(s_mem + 2 * P.nf);                // [nf] CurrentFrame.mvKeysUn[].angle
    unsigned* s_match = (unsigned*)(s_mem + 3 * P.nf);        // [nf] matches in the order they were made: idx2 | bin << 16
    const int2 pr = pairs[p];
    const uint8_t* recL = records + (size_t)pr.x * P.recBytes;
    const uint8_t* recC = records + (size_t)pr.y * P.recBytes;
    const int nL = rec_count(recL, P.nf), nC = rec_count(recC, P.nf);
    const ivf_keypoint* kc = rec_kps(recC);
    const ivf_keypoint* kl = rec_kps(recL);
    const float* urC = rec_uright(recC, P.nf);
    for (int i = lane; i < nC; i += 64) { s_assign[i] = -1; s_ur[i] = urC[i]; s_ang[i] = kc[i].angle; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Query* Q = queries + (size_t)p * P.nf;
    const int* C = count + (size_t)p * P.nf;
    const unsigned* Lp = lists + (size_t)p * P.nf * kListCap;
    int nm = 0, nMatch = 0, hist = 0;                          // hist: lane b counts rotHist[b].size()
    const float factor = 1.0f / 30.0f;                         // 1.0f / HISTO_LENGTH (:1380)

    for (int g0 = 0; g0 < nL; g0 += kPrefetch) {
        // the group's counts, query records, angles (lanes 0..15) and list entries (every lane, one per query) are requested
        // together, so their latencies overlap instead of adding up query by query
        const int gi = g0 + (lane & (kPrefetch - 1));
        const bool gv = gi < nL;
        const int myCnt = gv ? C[gi] : 0;
        Query myQ; myQ.u = 0; myQ.v = 0; myQ.ur = 0; myQ.bits = 0;
        float myAng = 0.0f;
        if (gv) { myQ = Q[gi]; myAng = kl[gi].angle; }
        unsigned ent[kPrefetch];
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) ent[k] = (g0 + k < nL) ? Lp[(size_t)(g0 + k) * kListCap + lane] : 0u;
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) {
            const int i = g0 + k;
            const int cnt = __builtin_amdgcn_readlane(myCnt, k);
            if (i >= nL || cnt == 0) continue;                                        // vIndices2.empty() (:1439)
            const float qur = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myQ.ur), k));
            const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)myQ.bits, k);
            const float radius = th * P.scale[bits & 0xff];
            int bestDist = 256, bestIdx2 = -1;
            if (cnt <= kListCap) {
                const unsigned e = ent[k];
                const int i2 = e & 0xffff, d = e >> 16;
                bool ok = lane < cnt;
                if (ok) {
                    const int a = s_assign[i2];
                    if (a >= 0 && !(a & kNoBlock)) ok = false;                         // occupied by a point with observations (:1447-1449)
                    const float u2 = s_ur[i2];
                    if (u2 > 0) { const float er = fabsf(qur - u2); if (er > radius) ok = false; }   // :1451-1457
                }
                const unsigned key = ok ? ((unsigned)d << 6) | (unsigned)lane : 0xffffffffu;
                const unsigned best = wave_min_u32_dpp(key);                            // first minimum in list order (:1463-1467)
                if (best != 0xffffffffu) {
                    bestDist = best >> 6;
                    bestIdx2 = __builtin_amdgcn_readlane(i2, best & 63);
                }
            } else {
                // the window overflowed its list: walk it again, against the live assignment state
                const float qu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myQ.u), k));
                const float qv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myQ.v), k));
                const int lo = (int)((bits >> 8) & 0xff) - 1, hi = (int)((bits >> 16) & 0xff) - 1;
                const uint4* qd = (const uint4*)(rec_desc(recL, P.nf) + (size_t)i * 32);
                const uint4 qa = qd[0], qb = qd[1];
                unsigned long long bestKey = ~0ull;                                    // dist << 32 | ordinal
                unsigned ordinal = 0;
                walk_window(P, kc, rec_desc(recC, P.nf), gStart + (size_t)p * (kGC * kGR + 1), gIdx + (size_t)p * P.nf, qu, qv, radius,
                            lo, hi, qa, qb, lane, [&](bool ok, int i2, int d) {
                                const unsigned long long m = __ballot(ok);
                                const unsigned pos = ordinal + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                                ordinal += (unsigned)__popcll(m);
                                if (ok) {
                                    const int a = s_assign[i2];
                                    if (a >= 0 && !(a & kNoBlock)) ok = false;
                                    const float u2 = s_ur[i2];
                                    if (u2 > 0) { const float er = fabsf(qur - u2); if (er > radius) ok = false; }
                                }
                                if (ok) {
                                    const unsigned long long key = ((unsigned long long)(unsigned)d << 32) | (unsigned long long)pos;
                                    if (key < bestKey) { bestKey = key; bestIdx2 = i2; }
                                }
                            });
                // lanes hold their own best (key, i2): reduce to the smallest key
                unsigned hiK = (unsigned)(bestKey >> 32), loK = (unsigned)bestKey;
                const unsigned minHi = wave_min_u32_dpp(hiK);
                const unsigned loC = hiK == minHi ? loK : 0xffffffffu;
                const unsigned minLo = wave_min_u32_dpp(loC);
                const unsigned long long who = __ballot(hiK == minHi && loK == minLo && bestKey != ~0ull);
                if (who) { bestDist = (int)minHi; bestIdx2 = __builtin_amdgcn_readlane(bestIdx2, __ffsll((long long)who) - 1); }
                else bestIdx2 = -1;
            }
            if (bestIdx2 >= 0 && bestDist <= 100) {                                     // TH_HIGH (:1469)
                const int blocks = (bits >> 25) & 1;
                int bin = 0;
                if (P.checkOri) {
                    const float qang = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myAng), k));
                    float rot = qang - s_ang[bestIdx2];                                 // :1476-1484
                    if (rot < 0.0f) rot += 360.0f;
                    bin = (int)roundf(rot * factor);
                    if (bin == 30) bin = 0;
                    hist += (lane == bin) ? 1 : 0;
                }
                if (lane == 0) {
                    s_assign[bestIdx2] = blocks ? i : (i | kNoBlock);
                    s_match[nMatch] = (unsigned)bestIdx2 | ((unsigned)bin << 16);
                }
                nMatch++; nm++;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    // ---- rotation consistency: ComputeThreeMaxima (:1654-1695) over the 30 bins, matches of every other bin are taken back
    if (P.checkOri) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < 30; b++) {
            const int s = __builtin_amdgcn_readlane(hist, b);
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
            else if (s > max3) { max3 = s; ind3 = b; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        __builtin_amdgcn_wave_barrier();
        int removed = 0;
        for (int m0 = 0; m0 < nMatch; m0 += 64) {
            const int m = m0 + lane;
            bool rm = false;
            if (m < nMatch) {
                const unsigned e = s_match[m];
                const int bin = (int)(e >> 16);
                rm = bin != ind1 && bin != ind2 && bin != ind3;
                if (rm) s_assign[e & 0xffff] = -1;                                      // :1504
            }
            removed += __popcll(__ballot(rm));
        }
        nm -= removed;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int* out = assignOut + (size_t)p * P.nf;
    for (int i = lane; i < P.nf; i += 64) { const int a = i < nC ? s_assign[i] : -1; out[i] = a < 0 ? -1 : (a & 0xffff); }
    if (lane == 0) {
        nmatchesOut[p] = nm;
        retryFlag[p] = (!onlyFlagged && nm < retryBelow) ? 1 : 0;                      // Tracking.cc:1320
    }
}

}  // namespace

// ---- C-ABI ---------------------------------------------------------------------------------------------------------------
struct ivf_tracker {
    ivf_track_config cfg;
    TrackParams P;
    int *dStart = nullptr, *dCount = nullptr, *dRetry = nullptr;
    unsigned short* dIdx = nullptr;
    Query* dQ = nullptr;
    unsigned* dLists = nullptr;
    size_t ldsBytes = 0;
    hipEvent_t evDone = nullptr;          // end of the previous run: the scratch above belongs to ONE run at a time
    bool ran = false;
};

extern "C" {

size_t ivf_track_record_bytes(int nfeatures)
{
    return nfeatures < 0 ? 0 : 16 + (size_t)nfeatures * (sizeof(ivf_keypoint) + 32 + 2 * sizeof(float));
}

void ivf_tracker_destroy(ivf_tracker* t)
{
    if (!t) return;
    (void)hipSetDevice(t->cfg.device_id);
    void* ptrs[] = {t->dStart, t->dCount, t->dRetry, t->dIdx, t->dQ, t->dLists};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (t->evDone) (void)hipEventDestroy(t->evDone);
    delete t;
}

int ivf_tracker_create(const ivf_track_config* cfg, ivf_tracker** out)
{
    if (!cfg || !out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->nfeatures < 1 || cfg->nfeatures > kMaxTrackFeatures)
        return fail(IVF_E_CAPACITY, "the batched tracker keeps 16 bytes of state per keypoint in LDS: nfeatures %d outside [1,%d]", cfg->nfeatures, kMaxTrackFeatures);
    if (cfg->nlevels < 1 || cfg->nlevels > kMaxLevels) return fail(IVF_E_INVALID, "nlevels %d outside [1,%d]", cfg->nlevels, kMaxLevels);
    if (cfg->max_pairs < 1) return fail(IVF_E_INVALID, "max_pairs must be >= 1");
    if (!(cfg->bounds.max_x > cfg->bounds.min_x) || !(cfg->bounds.max_y > cfg->bounds.min_y)) return fail(IVF_E_INVALID, "empty image bounds");
    if (!(cfg->fx > 0) || !(cfg->fy > 0) || !(cfg->bf > 0) || !(cfg->b > 0)) return fail(IVF_E_INVALID, "fx, fy, bf and b must be positive");
    if (!(cfg->th > 0) || (cfg->retry_below > 0 && !(cfg->th_retry > 0))) return fail(IVF_E_INVALID, "th (and th_retry with retry_below > 0) must be positive");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(IVF_E_NO_DEVICE, "no HIP device available; libivfront has no CPU path");
    if (cfg->device_id < 0 || cfg->device_id >= n) return fail(IVF_E_INVALID, "device_id %d outside [0,%d)", cfg->device_id, n);
    HIPCHK(hipSetDevice(cfg->device_id));
    ivf_tracker* t = new ivf_tracker();
    t->cfg = *cfg;
    TrackParams& P = t->P;
    memset(&P, 0, sizeof P);
    P.nf = cfg->nfeatures; P.nlevels = cfg->nlevels;
    for (int l = 0; l < kMaxLevels; l++) P.scale[l] = l < cfg->nlevels ? cfg->scale_factors[l] : 1.0f;
    P.fx = cfg->fx; P.fy = cfg->fy; P.cx = cfg->cx; P.cy = cfg->cy; P.invfx = 1.0f / cfg->fx; P.invfy = 1.0f / cfg->fy;   // Frame.cc:203-204
    P.bf = cfg->bf; P.b = cfg->b;
    P.minX = cfg->bounds.min_x; P.minY = cfg->bounds.min_y; P.maxX = cfg->bounds.max_x; P.maxY = cfg->bounds.max_y;
    P.invW = (float)kGC / (P.maxX - P.minX); P.invH = (float)kGR / (P.maxY - P.minY);                                    // Frame.cc:208-209
    P.thDepth = cfg->th_depth; P.checkOri = cfg->check_orientation ? 1 : 0; P.defaultBlocks = cfg->points_block ? 1 : 0;
    P.recBytes = ivf_track_record_bytes(cfg->nfeatures);
    const size_t np = (size_t)cfg->max_pairs, nf = (size_t)cfg->nfeatures;
    t->ldsBytes = nf * 16;
    if (hipMalloc(&t->dStart, np * (kGC * kGR + 1) * sizeof(int)) != hipSuccess || hipMalloc(&t->dIdx, np * nf * sizeof(unsigned short)) != hipSuccess ||
        hipMalloc(&t->dQ, np * nf * sizeof(Query)) != hipSuccess || hipMalloc(&t->dCount, np * nf * sizeof(int)) != hipSuccess ||
        hipMalloc(&t->dLists, np * nf * kListCap * sizeof(unsigned)) != hipSuccess || hipMalloc(&t->dRetry, np * sizeof(int)) != hipSuccess) {
        ivf_tracker_destroy(t);
        return fail(IVF_E_NO_DEVICE, "device allocation failed for a tracker of %d pairs x %d features", cfg->max_pairs, cfg->nfeatures);
    }
    if (hipEventCreateWithFlags(&t->evDone, hipEventDisableTiming) != hipSuccess) {
        ivf_tracker_destroy(t);
        return fail(IVF_E_NO_DEVICE, "event creation failed");
    }
    if (t->ldsBytes > 48 * 1024 &&
        hipFuncSetAttribute((const void*)k_track_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->ldsBytes) != hipSuccess) {
        ivf_tracker_destroy(t);
        return fail(IVF_E_NO_DEVICE, "cannot reserve %zu bytes of LDS for k_track_greedy", t->ldsBytes);
    }
    *out = t;
    return IVF_OK;
}

int ivf_tracker_run(ivf_tracker* t, const uint8_t* d_records, size_t record_bytes, int n_records, const int32_t* d_pairs, int n_pairs,
                    const float* d_poses, const uint8_t* d_point_flags, int32_t* d_assign, int32_t* d_nmatches, void* hip_stream)
{
    if (!t || !d_records || !d_pairs || !d_assign || !d_nmatches) return fail(IVF_E_INVALID, "null argument");
    if (record_bytes != t->P.recBytes) return fail(IVF_E_INVALID, "record_bytes %zu: records of %d features are %zu bytes", record_bytes, t->P.nf, t->P.recBytes);
    if (((size_t)d_records & 15) != 0) return fail(IVF_E_INVALID, "the record block must be 16-byte aligned");
    if (n_records < 1) return fail(IVF_E_INVALID, "n_records must be >= 1");
    if (n_pairs == 0) return IVF_OK;
    if (n_pairs < 0 || n_pairs > t->cfg.max_pairs) return fail(IVF_E_INVALID, "n_pairs %d outside [0,%d]", n_pairs, t->cfg.max_pairs);
    HIPCHK(hipSetDevice(t->cfg.device_id));
    hipStream_t st = (hipStream_t)hip_stream;
    const TrackParams& P = t->P;
    const int2* pairs = (const int2*)d_pairs;
    // the per-pair grids, query tables and candidate lists are scratch of the HANDLE: a run enqueued on another stream than the
    // previous one queues behind it (use one tracker per stream to let runs overlap)
    if (t->ran) HIPCHK(hipStreamWaitEvent(st, t->evDone, 0));
    hipLaunchKernelGGL(k_track_prepare, dim3(n_pairs), dim3(256), 0, st, P, d_records, pairs, d_poses, d_point_flags, t->dStart, t->dIdx, t->dQ, t->dRetry);
    const dim3 wg((P.nf + 3) / 4, n_pairs);
    for (int pass = 0; pass < (t->cfg.retry_below > 0 ? 2 : 1); pass++) {
        const float th = pass ? t->cfg.th_retry : t->cfg.th;
        hipLaunchKernelGGL(k_track_window, wg, dim3(256), 0, st, P, d_records, pairs, t->dStart, t->dIdx, t->dQ, th, t->dRetry, pass, t->dCount, t->dLists);
        hipLaunchKernelGGL(k_track_greedy, dim3(n_pairs), dim3(64), t->ldsBytes, st, P, d_records, pairs, t->dStart, t->dIdx, t->dQ, th, t->dRetry, pass,
                           t->cfg.retry_below, t->dCount, t->dLists, d_assign, d_nmatches);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(t->evDone, st));
    t->ran = true;
    return IVF_OK;
}

}  // extern "C"
