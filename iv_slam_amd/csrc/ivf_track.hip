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
// ivf_tracker_search_local is the second matcher call of a tracked frame, Tracking::SearchLocalPoints (Tracking.cc:2088-2132), for many
// frames at once: Frame::isInFrustum of every local map point (Frame.cc:557-613, MapPoint::PredictScale MapPoint.cc:407-422 with
// glibc's logf restated) and ORBmatcher::SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:45-135): windows
// on levels [level - 1, level], stereo check, best / second best in candidate order, the ratio test between candidates of the
// same octave, and the ORDER-DEPENDENT occupancy rule (a keypoint taken by a point with observations is skipped by later points).
//   k_local_prepare  one workgroup per frame: the same grid, then one thread per map point: projection -> query + radius
//   k_local_window   one wave per (frame, map point): ordered candidate list, entry = index | octave << 12 | distance << 16
//   k_local_greedy   one wave per frame: walks the points in order against the occupancy state in LDS; best and second best are
//                    two DPP minimum reductions of (distance << 6 | list position)
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
constexpr int kMaxTrackFeatures = 4096;           // LDS state of k_track_greedy: 20 B per keypoint

struct TrackParams {                               // uniform kernel arguments
    int nf, nlevels;
    float scale[kMaxLevels];
    float fx, fy, cx, cy, invfx, invfy, bf, b;
    float minX, minY, maxX, maxY, invW, invH;     // image bounds and mfGridElementWidthInv / HeightInv (Frame.cc:208-209)
    float thDepth;                                // > 0: UpdateLastFrame's close-point rule
    float logScale;                               // mfLogScaleFactor = logf(mfScaleFactor) (Frame.cc:106)
    int checkOri, defaultBlocks;
    int nRecords;                                 // records in the block of THIS call: every index read from a pair / frame table is checked against it
    size_t recBytes;
};
// a pair / frame table built for another block (a different world size or batch) must not read outside this one: such an entry
// gets nmatches = -1, assign = -1 and is otherwise skipped
DEVINL bool rec_ok(const TrackParams& P, int r) { return (unsigned)r < (unsigned)P.nRecords; }

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

// Frame::AssignFeaturesToGrid (Frame.cc:415-430) by one 256-thread workgroup: CSR, buckets in ix-major order, keypoints of a
// bucket in insertion order (rank inside the bucket = same-bucket keypoints in earlier 256-blocks + earlier threads).
// cnt [kGC * kGR], part [256], blk [256]: workgroup scratch in LDS.
DEVINL void build_grid(const TrackParams& P, const ivf_keypoint* __restrict__ kc, int nC, int* __restrict__ start,
                       unsigned short* __restrict__ idx, int* cnt, int* part, int* blk, int tid)
{
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
    if (!rec_ok(P, pr.x) || !rec_ok(P, pr.y)) { if (tid == 0) retryFlag[p] = 0; return; }
    const uint8_t* recL = records + (size_t)pr.x * P.recBytes;
    const uint8_t* recC = records + (size_t)pr.y * P.recBytes;
    const int nL = rec_count(recL, P.nf), nC = rec_count(recC, P.nf);
    if (tid == 0) { retryFlag[p] = 0; s_nStereo = 0; s_nClose = 0; }

    // ---- Frame::AssignFeaturesToGrid of the current frame (Frame.cc:415-430)
    const ivf_keypoint* kc = rec_kps(recC);
    build_grid(P, kc, nC, gStart + (size_t)p * (kGC * kGR + 1), gIdx + (size_t)p * P.nf, cnt, part, blk, tid);

    // ---- poses of THIS pair: {LastFrame.mTcw, CurrentFrame.mTcw = mVelocity * mLastFrame.mTcw (Tracking.cc:1311)}, row-major 3x4;
    //      identity when none are given (zero-motion prior)
    float Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tl[3] = {0, 0, 0}, Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tc[3] = {0, 0, 0};
    if (poses) {
        const float* Tl = poses + (size_t)p * 24; const float* Tc = Tl + 12;
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
    if (!rec_ok(P, pr.x) || !rec_ok(P, pr.y)) return;
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
                                                    const unsigned* __restrict__ lists, float* __restrict__ pointQuality,
                                                    float* __restrict__ keyQuality, int* __restrict__ assignOut,
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
    if (!rec_ok(P, pr.x) || !rec_ok(P, pr.y)) {
        for (int i = lane; i < P.nf; i += 64) assignOut[(size_t)p * P.nf + i] = -1;
        if (lane == 0) { nmatchesOut[p] = -1; retryFlag[p] = 0; }
        return;
    }
    const uint8_t* recL = records + (size_t)pr.x * P.recBytes;
    const uint8_t* recC = records + (size_t)pr.y * P.recBytes;
    const int nL = rec_count(recL, P.nf), nC = rec_count(recC, P.nf);
    const ivf_keypoint* kc = rec_kps(recC);
    const ivf_keypoint* kl = rec_kps(recL);
    const float* urC = rec_uright(recC, P.nf);
    for (int i = lane; i < nC; i += 64) { s_assign[i] = -1; s_ur[i] = urC[i]; s_ang[i] = kc[i].angle; }
    __shared__ float s_scale[kMaxLevels];
    __shared__ int s_hist[32];                                // rotHist[b].size()
    int* s_claim = s_mem + 4 * P.nf;                          // [nf] scratch of the group commit
    if (lane < kMaxLevels) s_scale[lane] = P.scale[lane];
    if (lane < 32) s_hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Query* Q = queries + (size_t)p * P.nf;
    const int* C = count + (size_t)p * P.nf;
    const unsigned* Lp = lists + (size_t)p * P.nf * kListCap;
    const float factor = 1.0f / 30.0f;                         // 1.0f / HISTO_LENGTH (:1380)

    // r04: a group of 16 queries is evaluated SPECULATIVELY against the live state -- four queries per pass, one per DPP row of 16 lanes
    // (a window holds a handful of candidates: radius 7 px x the octave's scale) -- and the longest PREFIX of the group in which no two
    // matched queries chose the same keypoint is committed at once: for those queries the state they would have seen in turn differs
    // from the evaluated one only by assignments to keypoints none of them chose, so the speculative first minimum is the serial one.
    // The first query that lost a keypoint to an earlier one of the group (the same corner found at two pyramid levels: common) starts
    // the next round, evaluated against the state that now holds the winner; a query whose list is longer than a row is walked alone
    // against the live state (64 entries per step, or its window again when the list overflowed).  The next group's lists are
    // requested before the current group is evaluated.
    // r06: the loads of a group are UNCONDITIONAL (index clamped into the frame's own arrays) and their results are touched for the first time when the group is
    // evaluated, two groups later; validity is re-derived from the indices there.  The r04 form -- `valid ? load : 0` per field, then `cur = nxt; nxt = nx2` --
    // compiled to branches around the loads with PHI copies behind them and register moves of the newest group at the loop's end: an s_waitcnt vmcnt(0) at the
    // head of EVERY group, i.e. no prefetch at all (phase timers: 56 % of the walk was that wait).  The three groups now rotate by unrolling, not by copies.
    struct Grp { unsigned ent[4]; int cnt[4]; float2 urb[4]; int myCnt; Query myQ; float myAng; };
    const int nLc = max(nL - 1, 0);
    auto load_group = [&](int g0, Grp& g) {
        const int gi = min(g0 + (lane & (kPrefetch - 1)), nLc);
        g.myCnt = C[gi]; g.myQ = Q[gi]; g.myAng = kl[gi].angle;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int qi = min(g0 + 4 * m + (lane >> 4), nLc);
            g.ent[m] = Lp[(size_t)qi * kListCap + (lane & 15)];
            g.cnt[m] = C[qi];
            g.urb[m] = *(const float2*)&Q[qi].ur;
        }
    };
#ifdef IVF_TRACK_TIMING
    const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
    int dbgRounds = 0, dbgBig = 0, dbgGroups = 0, dbgC4 = 0, dbgC8 = 0, dbgC0 = 0;
    // r06: where a round's time goes -- [0] group loads + loop head, [1] speculative evaluation (candidate tests, row minima, gathers: the part four waves
    // could share), [2] claim / commit (LDS claims, ballots, the prefix: serial by construction), [3] a long list walked alone, [4] the group's bins + match list
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tlast = tt0;
#define TRK_TIM(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TRK_TIM(i) do { } while (0)
#endif
    int nmAcc = 0, nMatchAcc = 0;
    auto process = [&](const int g0, const Grp& cur) {
        int nm = nmAcc, nMatch = nMatchAcc;
        int fb = -1;                              // lane k: the keypoint query k matched (-1: none)
        const int kEnd = min(kPrefetch, nL - g0);
        const int myCnt = (g0 + (lane & (kPrefetch - 1))) < nL ? cur.myCnt : 0;
        const Query myQ = cur.myQ;
        const unsigned bigMask = (unsigned)__ballot(lane < kEnd && myCnt > 16);
        int kstart = 0;
#ifdef IVF_TRACK_TIMING
        dbgGroups++;
        dbgC4 += __popcll(__ballot(lane < kEnd && myCnt > 4)); dbgC8 += __popcll(__ballot(lane < kEnd && myCnt > 8));
        dbgC0 += __popcll(__ballot(lane < kEnd && myCnt == 0));
#endif
        TRK_TIM(0);
        while (kstart < kEnd) {
#ifdef IVF_TRACK_TIMING
            if ((bigMask >> kstart) & 1u) dbgBig++; else dbgRounds++;
#endif
            if ((bigMask >> kstart) & 1u) {
                // ---- a long list: this query alone, against the live state
                const int k = kstart, i = g0 + k;
                const int cnt = __builtin_amdgcn_readlane(myCnt, k);
                const float qur = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myQ.ur), k));
                const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)myQ.bits, k);
                const float radius = th * P.scale[bits & 0xff];
                int bestDist = 256, bestIdx2 = -1;
                if (cnt <= kListCap) {
                    const unsigned e = Lp[(size_t)i * kListCap + lane];
                    const int i2 = e & 0xffff, d = e >> 16;
                    bool ok = lane < cnt;
                    if (ok) {
                        const int a = s_assign[i2];
                        if (a >= 0 && !(a & kNoBlock)) ok = false;                         // occupied by a point with observations (:1447-1449)
                        const float u2 = s_ur[i2];
                        if (u2 > 0) { const float er = fabsf(qur - u2); if (er > radius) ok = false; }   // :1451-1457
                    }
                    const unsigned key1 = ok ? ((unsigned)d << 6) | (unsigned)lane : 0xffffffffu;
                    const unsigned best = wave_min_u32_dpp(key1);                           // first minimum in list order (:1463-1467)
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
                                        const unsigned long long key2 = ((unsigned long long)(unsigned)d << 32) | (unsigned long long)pos;
                                        if (key2 < bestKey) { bestKey = key2; bestIdx2 = i2; }
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
                    if (lane == 0) s_assign[bestIdx2] = ((bits >> 25) & 1u) ? i : (i | kNoBlock);
                    if (lane == k) fb = bestIdx2;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                kstart++;
                TRK_TIM(3);
                continue;
            }
            // ---- speculative pass: row r of pass m = query 4 m + r
            unsigned key[4]; int i2m[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const unsigned e = cur.ent[m];
                const int i2 = e & 0xffff, d = e >> 16, pos = lane & 15;
                const int cntm = (g0 + 4 * m + (lane >> 4)) < nL ? cur.cnt[m] : 0;
                const float urm = cur.urb[m].x; const unsigned bitsm = __builtin_bit_cast(unsigned, cur.urb[m].y);
                bool ok = pos < cntm && cntm <= 16;
                if (ok) {
                    const int a = s_assign[i2];
                    if (a >= 0 && !(a & kNoBlock)) ok = false;
                    const float u2 = s_ur[i2];
                    if (u2 > 0) { const float er = fabsf(urm - u2); if (er > th * s_scale[bitsm & 0xff]) ok = false; }
                }
                i2m[m] = i2;
                key[m] = ok ? ((unsigned)d << 6) | (unsigned)pos : 0xffffffffu;
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {             // minimum of each row of 16 lanes, in its lane 15
                unsigned v = key[m], t;
                t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;
                t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;
                t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;
                t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;
                key[m] = v;
            }
            // lane k < 16 collects query k's result (row k & 3 of pass k >> 2)
            const int src = 16 * (lane & 3) + 15, mSel = (lane >> 2) & 3;
            unsigned kq[4];
#pragma unroll
            for (int m = 0; m < 4; m++) kq[m] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * src, (int)key[m]);
            const unsigned best = mSel == 0 ? kq[0] : (mSel == 1 ? kq[1] : (mSel == 2 ? kq[2] : kq[3]));
            const int src2 = 16 * (lane & 3) + (int)(best & 15u);
            int iq[4];
#pragma unroll
            for (int m = 0; m < 4; m++) iq[m] = __builtin_amdgcn_ds_bpermute(4 * src2, i2m[m]);
            const int bi = mSel == 0 ? iq[0] : (mSel == 1 ? iq[1] : (mSel == 2 ? iq[2] : iq[3]));
            const bool mine = lane >= kstart && lane < kEnd;
            const bool mt = mine && myCnt != 0 && myCnt <= 16 && best != 0xffffffffu && (int)(best >> 6) <= 100;      // TH_HIGH (:1469)
            TRK_TIM(1);
            // the lowest query of every set that chose the same keypoint wins it; the others are losers
            if (mt) s_claim[bi] = 64;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (mt) atomicMin(&s_claim[bi], lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool loser = mt && s_claim[bi] != lane;
            const unsigned stopMask = ((unsigned)__ballot(loser) | bigMask) & (0xffffffffu << kstart);
            const int kstop = stopMask ? min(__ffs((int)stopMask) - 1, kEnd) : kEnd;
            if (mt && lane < kstop) {
                s_assign[bi] = ((myQ.bits >> 25) & 1u) ? (g0 + lane) : ((g0 + lane) | kNoBlock);
                fb = bi;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            kstart = kstop;
            TRK_TIM(2);
        }
        // ---- the group's matches: rotation bins (:1476-1484) and the match list, lane k = query k
        {
            const bool mt = fb >= 0;
            int bin = 0;
            if (P.checkOri && mt) {
                float rot = cur.myAng - s_ang[fb];
                if (rot < 0.0f) rot += 360.0f;
                bin = (int)roundf(rot * factor);
                if (bin == 30) bin = 0;
            }
            const unsigned long long mm = __ballot(mt);
            if (mt) s_match[nMatch + __popcll(mm & ((1ull << lane) - 1ull))] = (unsigned)fb | ((unsigned)bin << 16);
            if (P.checkOri && mt) atomicAdd(&s_hist[bin], 1);
            const int n = __popcll(mm);
            nMatch += n; nm += n;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        nmAcc = nm; nMatchAcc = nMatch;
        TRK_TIM(4);
    };
    {
        Grp gA, gB, gC;                       // two groups ahead: a group's work (~1 us) is shorter than a trip to L2 / HBM under load
        load_group(0, gA);
        load_group(kPrefetch, gB);
        for (int g0 = 0; g0 < nL; g0 += 3 * kPrefetch) {
            load_group(g0 + 2 * kPrefetch, gC); process(g0, gA);
            if (g0 + kPrefetch < nL) { load_group(g0 + 3 * kPrefetch, gA); process(g0 + kPrefetch, gB); }
            if (g0 + 2 * kPrefetch < nL) { load_group(g0 + 4 * kPrefetch, gB); process(g0 + 2 * kPrefetch, gC); }
        }
    }
    int nm = nmAcc, nMatch = nMatchAcc;
#ifdef IVF_TRACK_TIMING
    const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- rotation consistency: ComputeThreeMaxima (:1654-1695) over the 30 bins, matches of every other bin are taken back
    if (P.checkOri) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < 30; b++) {
            const int s = s_hist[b];
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
    if (pointQuality && keyQuality) {
        // ORBmatcher::UpdateQualityScores(CurrentFrame) at the end of EVERY SearchByProjection call (:1108-1121, :1513-1515; the
        // retry is a second call and updates again from what the first left).  A last-frame point ends on at most one current
        // keypoint, so the sequential loop of the reference has no cross-iteration dependence here: one lane per keypoint.
        float* pq = pointQuality + (size_t)p * P.nf;
        float* kq = keyQuality + (size_t)p * P.nf;
        for (int i = lane; i < nC; i += 64) {
            const int a = s_assign[i];
            if (a < 0) continue;
            const int m = a & 0xffff;
            const float mq = pq[m], upd = fminf(mq, kq[i]);
            if (fabsf(upd - mq) > 0.01f) pq[m] = upd;                                  // kDeltaThresh
            kq[i] = upd;
        }
    }
    if (lane == 0) {
        nmatchesOut[p] = nm;
        retryFlag[p] = (!onlyFlagged && nm < retryBelow) ? 1 : 0;                      // Tracking.cc:1320
    }
#ifdef IVF_TRACK_TIMING
    if (lane == 0 && (p == 1 || p == 40)) {
        const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
        printf("pair %d pass %d: loop %llu epilogue %llu cycles groups %d rounds %d big(>16) %d nm %d | queries with no candidate %d, more than 4: %d, more than 8: %d\n", p, onlyFlagged, tt1 - tt0, tt2 - tt1, dbgGroups, dbgRounds, dbgBig, nm, dbgC0, dbgC4, dbgC8);
        printf("pair %d pass %d: of the loop: group loads + head %llu, speculative evaluation %llu, claim / commit %llu, long lists %llu, bins + match list %llu (100 MHz ticks)\n", p, onlyFlagged,
               tph[0], tph[1], tph[2], tph[3], tph[4]);
    }
#endif
}


// ================================================================================================
// Tracking::SearchLocalPoints, batched (ivf_tracker_search_local)
// ================================================================================================
// glibc >= 2.27 logf (oracle/ivf_oracle.c:orc_logf is the same arithmetic, checked against libm for every positive
// normal float): what MapPoint::PredictScale's log(ratio) resolves to.  Positive normal arguments only (a distance ratio).
__constant__ double c_logfT[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
DEVINL float glibc_logf(float x)
{
    const unsigned ix = __builtin_bit_cast(unsigned, x);
    if (ix == 0x3f800000u) return 0.0f;
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int)tmp >> 23;
    const double z = (double)__builtin_bit_cast(float, ix - (tmp & 0xff800000u)), invc = c_logfT[i][0], logc = c_logfT[i][1];
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

static_assert(sizeof(ivf_local_point) == 80, "ivf_local_point is read as five 16-byte pieces");
constexpr int kLocalNoBlock = 0x40000000;         // assignment made by a point without observations: later points may replace it (:87-89)

// one workgroup per frame: grid, then Frame::isInFrustum + the window radius of every local map point
__global__ __launch_bounds__(256) void k_local_prepare(TrackParams P, const uint8_t* __restrict__ records, const int* __restrict__ frames,
                                                      const float* __restrict__ poses, const ivf_local_point* __restrict__ points,
                                                      const int* __restrict__ offsets, float th, float cosLimit, int maxM,
                                                      int* __restrict__ gStart, unsigned short* __restrict__ gIdx,
                                                      Query* __restrict__ queries, float* __restrict__ radii)
{
    __shared__ int cnt[kGC * kGR];
    __shared__ int part[256];
    __shared__ int blk[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int ri = frames[f];
    if (!rec_ok(P, ri) || offsets[f + 1] < offsets[f] || offsets[f] < 0) return;
    const uint8_t* recC = records + (size_t)ri * P.recBytes;
    const int nC = rec_count(recC, P.nf);
    build_grid(P, rec_kps(recC), nC, gStart + (size_t)f * (kGC * kGR + 1), gIdx + (size_t)f * P.nf, cnt, part, blk, tid);

    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, Ow[3];
    if (poses) {
        const float* T = poses + (size_t)f * 12;                                          // the pose of frame SLOT f (a record may appear in two slots)
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j];
            t[i] = T[4 * i + 3];
        }
    }
    neg_rt_mul(R, t, Ow);                                                             // mOw (Frame.cc:554)
    const int m0 = offsets[f], M = min(offsets[f + 1] - m0, maxM);
    Query* Q = queries + (size_t)f * maxM;
    float* Rd = radii + (size_t)f * maxM;
    for (int m = tid; m < M; m += 256) {
        const ivf_local_point mp = points[m0 + m];
        Query q; q.u = 0; q.v = 0; q.ur = 0; q.bits = 0;
        float radius = 0.0f;
        if (!(mp.flags & 1)) {                                                        // isBad() / mnLastFrameSeen == mnId (Tracking.cc:2114-2115)
            float Pc[3];
            mul_add(R, mp.pos, t, Pc);                                                // Frame.cc:565
            if (!(Pc[2] < 0.0f)) {                                                    // :571
                const float invz = 1.0f / Pc[2];
                const float u = P.fx * Pc[0] * invz + P.cx, v = P.fy * Pc[1] * invz + P.cy;
                if (!(u < P.minX || u > P.maxX) && !(v < P.minY || v > P.maxY)) {     // :579-582
                    const float PO[3] = {mp.pos[0] - Ow[0], mp.pos[1] - Ow[1], mp.pos[2] - Ow[2]};
                    const double n2 = (double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2];
                    const float dist = (float)sqrt(n2);                               // cv::norm (:588)
                    const float minD = 0.8f * mp.min_distance, maxD = 1.2f * mp.max_distance;     // MapPoint.cc:378-388
                    if (!(dist < minD || dist > maxD)) {                              // :590
                        const double dt = (double)PO[0] * (double)mp.normal[0] + (double)PO[1] * (double)mp.normal[1] + (double)PO[2] * (double)mp.normal[2];
                        const float viewCos = (float)(dt / (double)dist);             // :596
                        if (!(viewCos < cosLimit)) {                                  // :598
                            // MapPoint::PredictScale (MapPoint.cc:407-422)
                            const float ratio = mp.max_distance / dist;
                            int lv = (int)ceilf(glibc_logf(ratio) / P.logScale);
                            lv = lv < 0 ? 0 : (lv >= P.nlevels ? P.nlevels - 1 : lv);
                            float r = ((double)viewCos > 0.998) ? 2.5f : 4.0f;        // RadiusByViewingCos (ORBmatcher.cc:137-143)
                            if (th != 1.0f) r *= th;                                  // :65-66
                            radius = r * P.scale[lv];                                 // :69
                            q.u = u; q.v = v; q.ur = u - P.bf * invz;                 // :606-608
                            q.bits = (unsigned)lv | (1u << 24) | ((unsigned)((mp.flags >> 1) & 1) << 25);
                        }
                    }
                }
            }
        }
        Q[m] = q; Rd[m] = radius;
    }
}

// one wave per (frame, map point): the ordered candidate list of ORBmatcher.cc:68-100 minus the occupancy test
__global__ __launch_bounds__(256) void k_local_window(TrackParams P, const uint8_t* __restrict__ records, const int* __restrict__ frames,
                                                     const ivf_local_point* __restrict__ points, const int* __restrict__ offsets, int maxM,
                                                     const int* __restrict__ gStart, const unsigned short* __restrict__ gIdx,
                                                     const Query* __restrict__ queries, const float* __restrict__ radii,
                                                     int* __restrict__ count, unsigned* __restrict__ lists)
{
    const int f = blockIdx.y;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m0 = offsets[f], M = min(offsets[f + 1] - m0, maxM);
    if (m >= M || m0 < 0 || !rec_ok(P, frames[f])) return;
    const uint8_t* recC = records + (size_t)frames[f] * P.recBytes;
    const Query q = queries[(size_t)f * maxM + m];
    int total = 0;
    if ((q.bits >> 24) & 1) {
        const int lv = q.bits & 0xff;
        const float r = radii[(size_t)f * maxM + m];
        const uint4* qd = (const uint4*)points[m0 + m].desc;
        const uint4 qa = qd[0], qb = qd[1];
        const float* urC = rec_uright(recC, P.nf);
        const ivf_keypoint* kc = rec_kps(recC);
        unsigned* L = lists + ((size_t)f * maxM + m) * kListCap;
        walk_window(P, kc, rec_desc(recC, P.nf), gStart + (size_t)f * (kGC * kGR + 1), gIdx + (size_t)f * P.nf,
                    q.u, q.v, r, lv - 1, lv, qa, qb, lane, [&](bool ok, int i2, int d) {
                        if (ok) { const float u2 = urC[i2]; if (u2 > 0) { const float er = fabsf(q.ur - u2); if (er > r) ok = false; } }   // :91-96
                        const unsigned long long mk = __ballot(ok);
                        if (ok) {
                            const int pos = total + __popcll(mk & ((1ull << lane) - 1ull));
                            if (pos < kListCap) L[pos] = (unsigned)i2 | ((unsigned)kc[i2].octave << 12) | ((unsigned)d << 16);
                        }
                        total += __popcll(mk);
                    });
    }
    if (lane == 0) count[(size_t)f * maxM + m] = total;
}

// top two of (key = distance << 12 | ordinal) pairs: a (best, second) pair merged with another one
DEVINL void merge_top2(unsigned& kb, unsigned& eb, unsigned& ks, unsigned& es, unsigned k1, unsigned e1, unsigned k2, unsigned e2)
{
    if (k1 < kb) {
        if (kb < k2) { ks = kb; es = eb; } else { ks = k2; es = e2; }
        kb = k1; eb = e1;
    } else if (k1 < ks) { ks = k1; es = e1; }
}

// one wave per frame: the greedy walk of ORBmatcher.cc:51-126 over the frame's map points.
// Best / second best (:102-114) are updated sequentially in the reference: `dist < bestDist` demotes the best to second, otherwise
// `dist < bestDist2` replaces the second.  That equals "first and second of the candidates ordered by (distance, position)":
//   * the best is the first candidate of minimal distance (a later equal one fails the strict `<`);
//   * a demoted best was the (distance, position)-minimum of everything before the new best, a directly inserted second is the first
//     of its distance among the non-best seen so far, and a later candidate of the same distance never replaces either (strict `<`):
//     by induction the second is the (distance, position)-minimum of all candidates but the best.
// So two minimum reductions of (distance << 12 | position) -- the second with the winner's lane masked -- give both, with their octaves.
__global__ __launch_bounds__(64) void k_local_greedy(TrackParams P, const uint8_t* __restrict__ records, const int* __restrict__ frames,
                                                    const ivf_local_point* __restrict__ points, const int* __restrict__ offsets, int maxM,
                                                    const int* __restrict__ gStart, const unsigned short* __restrict__ gIdx,
                                                    const Query* __restrict__ queries, const float* __restrict__ radii,
                                                    const uint8_t* __restrict__ occupied, float nnRatio, const int* __restrict__ count,
                                                    const unsigned* __restrict__ lists, float* __restrict__ pointQuality,
                                                    float* __restrict__ keyQuality, int* __restrict__ assignOut,
                                                    int* __restrict__ nmatchesOut)
{
    extern __shared__ int s_mem[];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (!rec_ok(P, frames[f]) || offsets[f + 1] < offsets[f] || offsets[f] < 0) {
        for (int i = lane; i < P.nf; i += 64) assignOut[(size_t)f * P.nf + i] = -1;
        if (lane == 0) nmatchesOut[f] = -1;
        return;
    }
    int* s_assign = s_mem;                                    // [nf]: -1 free, -2 held by a point with observations before the call,
                                                              // >= 0 local point index (| kLocalNoBlock) assigned by this call
    const uint8_t* recC = records + (size_t)frames[f] * P.recBytes;
    const int nC = rec_count(recC, P.nf);
    const int m0 = offsets[f], M = min(offsets[f + 1] - m0, maxM);
    const uint8_t* occ = occupied ? occupied + (size_t)f * P.nf : nullptr;
    for (int i = lane; i < nC; i += 64) s_assign[i] = (occ && occ[i]) ? -2 : -1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Query* Q = queries + (size_t)f * maxM;
    const float* Rd = radii + (size_t)f * maxM;
    const int* C = count + (size_t)f * maxM;
    const unsigned* Lp = lists + (size_t)f * maxM * kListCap;
    int nm = 0;
    auto blocked = [&](int i2) { const int a = s_assign[i2]; return a == -2 || (a >= 0 && !(a & kLocalNoBlock)); };   // :87-89

    // r06: the next group's lists are in flight while this one is walked (unconditional loads at clamped indices, two register sets that alternate
    // by unrolling -- see k_track_greedy); before, every group of 16 queries started with a full trip to memory
    struct LGrp { unsigned ent[kPrefetch]; int myCnt; unsigned myBits; };
    const int Mc = max(M - 1, 0);
    auto load_group = [&](int g0, LGrp& g) {
        const int gi = min(g0 + (lane & (kPrefetch - 1)), Mc);
        g.myCnt = C[gi]; g.myBits = Q[gi].bits;
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) g.ent[k] = Lp[(size_t)min(g0 + k, Mc) * kListCap + lane];
    };
    auto process = [&](const int g0, const LGrp& grp) {
        const int myCnt = (g0 + (lane & (kPrefetch - 1))) < M ? grp.myCnt : 0;
        const unsigned myBits = grp.myBits;
        const unsigned (&ent)[kPrefetch] = grp.ent;
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) {
            const int m = g0 + k;
            const int cnt = __builtin_amdgcn_readlane(myCnt, k);
            if (m >= M || cnt == 0) continue;                                         // vIndices.empty() (:71)
            const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)myBits, k);
            unsigned kb = 0xffffffffu, eb = 0, ks = 0xffffffffu, es = 0;               // best / second: key = dist << 12 | ordinal, entry
            if (cnt <= kListCap) {
                const unsigned e = ent[k];
                const bool ok = lane < cnt && !blocked((int)(e & 0xfff));
                const unsigned key = ok ? ((e >> 16) << 12) | (unsigned)lane : 0xffffffffu;
                kb = wave_min_u32_dpp(key);
                if (kb != 0xffffffffu) {
                    const int bl = (int)(kb & 63);
                    eb = (unsigned)__builtin_amdgcn_readlane((int)e, bl);
                    ks = wave_min_u32_dpp(lane == bl ? 0xffffffffu : key);
                    if (ks != 0xffffffffu) es = (unsigned)__builtin_amdgcn_readlane((int)e, (int)(ks & 63));
                }
            } else {
                // the window overflowed its list: walk it again against the live occupancy state, 64 candidates at a time
                const Query q = Q[m];
                const float r = Rd[m];
                const int lv = bits & 0xff;
                const uint4* qd = (const uint4*)points[m0 + m].desc;
                const uint4 qa = qd[0], qb = qd[1];
                const float* urC = rec_uright(recC, P.nf);
                const ivf_keypoint* kc = rec_kps(recC);
                unsigned ordinal = 0;
                walk_window(P, kc, rec_desc(recC, P.nf), gStart + (size_t)f * (kGC * kGR + 1), gIdx + (size_t)f * P.nf, q.u, q.v, r,
                            lv - 1, lv, qa, qb, lane, [&](bool ok, int i2, int d) {
                                if (ok) { const float u2 = urC[i2]; if (u2 > 0) { const float er = fabsf(q.ur - u2); if (er > r) ok = false; } }
                                const unsigned long long mk = __ballot(ok);
                                const unsigned pos = ordinal + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
                                ordinal += (unsigned)__popcll(mk);
                                if (ok && blocked(i2)) ok = false;
                                const unsigned e = ok ? ((unsigned)i2 | ((unsigned)kc[i2].octave << 12) | ((unsigned)d << 16)) : 0u;
                                const unsigned key = ok ? ((unsigned)d << 12) | pos : 0xffffffffu;
                                const unsigned k1 = wave_min_u32_dpp(key);
                                if (k1 == 0xffffffffu) return;
                                const unsigned long long w1 = __ballot(key == k1);
                                const int l1 = __ffsll((long long)w1) - 1;
                                const unsigned e1 = (unsigned)__builtin_amdgcn_readlane((int)e, l1);
                                const unsigned k2 = wave_min_u32_dpp(lane == l1 ? 0xffffffffu : key);
                                unsigned e2 = 0;
                                if (k2 != 0xffffffffu) {
                                    const unsigned long long w2 = __ballot(key == k2 && lane != l1);
                                    e2 = (unsigned)__builtin_amdgcn_readlane((int)e, __ffsll((long long)w2) - 1);
                                }
                                merge_top2(kb, eb, ks, es, k1, e1, k2, e2);
                            });
            }
            if (kb == 0xffffffffu) continue;
            const int bestDist = (int)(kb >> 12), bestIdx = (int)(eb & 0xfff), bestLevel = (int)((eb >> 12) & 15);
            if (bestDist > 100) continue;                                               // TH_HIGH (:118)
            if (ks != 0xffffffffu) {
                const int bestDist2 = (int)(ks >> 12), bestLevel2 = (int)((es >> 12) & 15);
                if (bestLevel == bestLevel2 && (float)bestDist > nnRatio * (float)bestDist2) continue;   // :120-121
            }
            if (lane == 0) s_assign[bestIdx] = ((bits >> 25) & 1) ? m : (m | kLocalNoBlock);          // :123
            nm++;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };
    if (M > 0) {
        LGrp gA, gB;
        load_group(0, gA);
        for (int g0 = 0; g0 < M; g0 += 2 * kPrefetch) {
            load_group(g0 + kPrefetch, gB); process(g0, gA);
            if (g0 + kPrefetch < M) { load_group(g0 + 2 * kPrefetch, gA); process(g0 + kPrefetch, gB); }
        }
    }
    __builtin_amdgcn_wave_barrier();
    int* out = assignOut + (size_t)f * P.nf;
    for (int i = lane; i < P.nf; i += 64) { const int a = i < nC ? s_assign[i] : -1; out[i] = a < 0 ? -1 : (a & ~kLocalNoBlock); }
    if (pointQuality && keyQuality) {
        // UpdateQualityScores(F) (:128-132, :1108-1121) for the keypoints that received a point in THIS call (a map point ends on at
        // most one keypoint: no cross-iteration dependence).  Keypoints that held a point before the call went through the same
        // update at the end of the search that gave it to them, and the update is idempotent: min(q_mp, q_kp) == q_kp afterwards.
        float* pq = pointQuality + m0;
        float* kq = keyQuality + (size_t)f * P.nf;
        for (int i = lane; i < nC; i += 64) {
            const int a = s_assign[i];
            if (a < 0) continue;
            const int m = a & ~kLocalNoBlock;
            const float mq = pq[m], upd = fminf(mq, kq[i]);
            if (fabsf(upd - mq) > 0.01f) pq[m] = upd;
            kq[i] = upd;
        }
    }
    if (lane == 0) nmatchesOut[f] = nm;
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
    // scratch of ivf_tracker_search_local, sized at its first call / grown on demand: [max_pairs][localCap] queries, radii, counts, lists
    Query* dLQ = nullptr; float* dLRad = nullptr; int* dLCount = nullptr; unsigned* dLLists = nullptr; int localCap = 0;
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
    void* ptrs[] = {t->dStart, t->dCount, t->dRetry, t->dIdx, t->dQ, t->dLists, t->dLQ, t->dLRad, t->dLCount, t->dLLists};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (t->evDone) (void)hipEventDestroy(t->evDone);
    delete t;
}

int ivf_tracker_create(const ivf_track_config* cfg, ivf_tracker** out)
{
    if (!cfg || !out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->nfeatures < 1 || cfg->nfeatures > kMaxTrackFeatures)
        return fail(IVF_E_CAPACITY, "the batched tracker keeps 20 bytes of state per keypoint in LDS: nfeatures %d outside [1,%d]", cfg->nfeatures, kMaxTrackFeatures);
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
    P.logScale = cfg->nlevels > 1 ? logf(cfg->scale_factors[1]) : 1.0f;             // mfLogScaleFactor = log(mfScaleFactor) (Frame.cc:106): logf
    const size_t np = (size_t)cfg->max_pairs, nf = (size_t)cfg->nfeatures;
    t->ldsBytes = nf * 20;
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
                    const float* d_poses, const uint8_t* d_point_flags, float* d_point_quality, float* d_key_quality,
                    int32_t* d_assign, int32_t* d_nmatches, void* hip_stream)
{
    if (!t || !d_records || !d_pairs || !d_assign || !d_nmatches) return fail(IVF_E_INVALID, "null argument");
    if (record_bytes != t->P.recBytes) return fail(IVF_E_INVALID, "record_bytes %zu: records of %d features are %zu bytes", record_bytes, t->P.nf, t->P.recBytes);
    if (((size_t)d_records & 15) != 0) return fail(IVF_E_INVALID, "the record block must be 16-byte aligned");
    if (n_records < 1) return fail(IVF_E_INVALID, "n_records must be >= 1");
    if (n_pairs == 0) return IVF_OK;
    if (n_pairs < 0 || n_pairs > t->cfg.max_pairs) return fail(IVF_E_INVALID, "n_pairs %d outside [0,%d]", n_pairs, t->cfg.max_pairs);
    HIPCHK(hipSetDevice(t->cfg.device_id));
    if ((d_point_quality == nullptr) != (d_key_quality == nullptr)) return fail(IVF_E_INVALID, "d_point_quality and d_key_quality come together");
    hipStream_t st = (hipStream_t)hip_stream;
    TrackParams P = t->P;
    P.nRecords = n_records;
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
                           t->cfg.retry_below, t->dCount, t->dLists, d_point_quality, d_key_quality, d_assign, d_nmatches);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(t->evDone, st));
    t->ran = true;
    return IVF_OK;
}

int ivf_tracker_search_local(ivf_tracker* t, const uint8_t* d_records, size_t record_bytes, int n_records, const int32_t* d_frames, int n_frames,
                             const float* d_poses, const ivf_local_point* d_points, const int32_t* d_point_offsets, int max_points_per_frame,
                             const uint8_t* d_occupied, float th, float nn_ratio, float cos_limit, float* d_point_quality, float* d_key_quality,
                             int32_t* d_assign, int32_t* d_nmatches, void* hip_stream)
{
    if (!t || !d_records || !d_frames || !d_points || !d_point_offsets || !d_assign || !d_nmatches) return fail(IVF_E_INVALID, "null argument");
    if (record_bytes != t->P.recBytes) return fail(IVF_E_INVALID, "record_bytes %zu: records of %d features are %zu bytes", record_bytes, t->P.nf, t->P.recBytes);
    if (((size_t)d_records & 15) != 0 || ((size_t)d_points & 15) != 0) return fail(IVF_E_INVALID, "the record block and the point array must be 16-byte aligned");
    if (n_records < 1) return fail(IVF_E_INVALID, "n_records must be >= 1");
    if (n_frames == 0) return IVF_OK;
    if (n_frames < 0 || n_frames > t->cfg.max_pairs) return fail(IVF_E_INVALID, "n_frames %d outside [0,%d]", n_frames, t->cfg.max_pairs);
    if (max_points_per_frame < 1 || max_points_per_frame >= kLocalNoBlock) return fail(IVF_E_INVALID, "max_points_per_frame %d out of range", max_points_per_frame);
    if (!(th > 0) || !(nn_ratio > 0)) return fail(IVF_E_INVALID, "th and nn_ratio must be positive");
    if ((d_point_quality == nullptr) != (d_key_quality == nullptr)) return fail(IVF_E_INVALID, "d_point_quality and d_key_quality come together");
    HIPCHK(hipSetDevice(t->cfg.device_id));
    hipStream_t st = (hipStream_t)hip_stream;
    if (t->ran) HIPCHK(hipStreamWaitEvent(st, t->evDone, 0));                          // the handle's scratch belongs to one call at a time
    if (max_points_per_frame > t->localCap) {
        // grow the scratch: the previous call (if any) must be done with the old buffers before they are freed
        if (t->ran) HIPCHK(hipEventSynchronize(t->evDone));
        void* old[] = {t->dLQ, t->dLRad, t->dLCount, t->dLLists};
        for (void* q : old) if (q) (void)hipFree(q);
        t->dLQ = nullptr; t->dLRad = nullptr; t->dLCount = nullptr; t->dLLists = nullptr; t->localCap = 0;
        const size_t n = (size_t)t->cfg.max_pairs * (size_t)max_points_per_frame;
        if (hipMalloc(&t->dLQ, n * sizeof(Query)) != hipSuccess || hipMalloc(&t->dLRad, n * sizeof(float)) != hipSuccess ||
            hipMalloc(&t->dLCount, n * sizeof(int)) != hipSuccess || hipMalloc(&t->dLLists, n * kListCap * sizeof(unsigned)) != hipSuccess) {
            void* part[] = {t->dLQ, t->dLRad, t->dLCount, t->dLLists};
            for (void* q : part) if (q) (void)hipFree(q);
            t->dLQ = nullptr; t->dLRad = nullptr; t->dLCount = nullptr; t->dLLists = nullptr;
            return fail(IVF_E_NO_DEVICE, "device allocation failed for %d frames x %d local map points", t->cfg.max_pairs, max_points_per_frame);
        }
        t->localCap = max_points_per_frame;
    }
    TrackParams P = t->P;
    P.nRecords = n_records;
    const int M = max_points_per_frame;
    hipLaunchKernelGGL(k_local_prepare, dim3(n_frames), dim3(256), 0, st, P, d_records, d_frames, d_poses, d_points, d_point_offsets, th, cos_limit, M,
                       t->dStart, t->dIdx, t->dLQ, t->dLRad);
    hipLaunchKernelGGL(k_local_window, dim3((M + 3) / 4, n_frames), dim3(256), 0, st, P, d_records, d_frames, d_points, d_point_offsets, M,
                       t->dStart, t->dIdx, t->dLQ, t->dLRad, t->dLCount, t->dLLists);
    hipLaunchKernelGGL(k_local_greedy, dim3(n_frames), dim3(64), (size_t)P.nf * 4, st, P, d_records, d_frames, d_points, d_point_offsets, M,
                       t->dStart, t->dIdx, t->dLQ, t->dLRad, d_occupied, nn_ratio, t->dLCount, t->dLLists, d_point_quality, d_key_quality,
                       d_assign, d_nmatches);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(t->evDone, st));
    t->ran = true;
    return IVF_OK;
}

}  // extern "C"
