// ivf_api.hip -- C-ABI of libivfront.so (include/ivfront.h): handles, geometry, launch orchestration.
// Host side only; all image / descriptor compute is in ivf_kernels.hip.  No CPU fallback exists.
#include "ivf_device.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include <algorithm>

using namespace ivf;

#ifdef IVF_EXPERIMENT
// ---- soak aid (tools/soak_frontend.py): host-visible progress words ----
// A 1-thread marker launch behind every stage of Context::run / ivf_frontend_run writes (run << 8 | stage) into a word of
// host-mapped memory the caller registered with ivf_debug_progress_words: when a child process of the soak sticks in
// ivf_frontend_sync, its parent reads from these words which stage each internal stream reached.  Words:
//   [k]      own batch on internal stream k:  1 ingest  2 pyramid  3 FAST  4 selection  5 join (blur awaited)  6 descriptors  7 stereo
//   [3 + k]  a LENT blur on internal stream k (batch of context k - 1): 1 fork passed  2 blur done
//   [6]      host: run << 8 | 1 enqueue entered, 2 enqueue done; [7] host: 0x10 + k = ivf_frontend_sync waits for stream k, 0x20 = returned
// Experiment builds only: the product has neither the words nor the launches.
namespace ivf {
__global__ void k_mark(int* w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static int* g_markDev = nullptr;
static volatile int* g_markHost = nullptr;
#define IVF_MARK(st, word, stage, run) \
    do { if (g_markDev && (word) >= 0) hipLaunchKernelGGL(ivf::k_mark, dim3(1), dim3(1), 0, (st), g_markDev + (word), (int)(((run) << 8) | (stage))); } while (0)
#define IVF_MARK_HOST(word, v) do { if (g_markHost) g_markHost[(word)] = (int)(v); } while (0)
#else
#define IVF_MARK(st, word, stage, run) do { } while (0)
#define IVF_MARK_HOST(word, v) do { } while (0)
#endif

namespace {

thread_local std::string g_err;

#define fail ivf::set_error
#define HIPCHK(expr)                                                                                   \
    do { hipError_t e_ = (expr);                                                                        \
         if (e_ != hipSuccess) return fail(IVF_E_NO_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// ---- ORBextractor constructor tables (ORB/src/ORBextractor.cc:411-476) ----
struct Tables {
    ivf_extractor_params p;
    float scale[kMaxLevels], invScale[kMaxLevels], sigma2[kMaxLevels], invSigma2[kMaxLevels];
    int nfeat[kMaxLevels];
    int umax[16];
};
int make_tables(const ivf_extractor_params& p, Tables& t)
{
    if (p.nlevels < 1 || p.nlevels > kMaxLevels) return fail(IVF_E_INVALID, "nlevels %d outside [1,%d]", p.nlevels, kMaxLevels);
    if (p.nfeatures < 1 || p.nfeatures > 65534) return fail(IVF_E_INVALID, "nfeatures %d outside [1,65534]", p.nfeatures);
    if (!(p.scale_factor > 1.0f)) return fail(IVF_E_INVALID, "scale_factor must be > 1");
    if (p.min_th_fast < 1 || p.ini_th_fast < p.min_th_fast || p.ini_th_fast > 254)
        return fail(IVF_E_INVALID, "need 1 <= minThFAST <= iniThFAST <= 254");
    t.p = p;
    const double sf = (double)p.scale_factor;            // member is `double scaleFactor` (ORBextractor.h:110)
    t.scale[0] = 1.0f; t.sigma2[0] = 1.0f;
    for (int i = 1; i < p.nlevels; i++) {
        t.scale[i] = (float)((double)t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < p.nlevels; i++) { t.invScale[i] = 1.0f / t.scale[i]; t.invSigma2[i] = 1.0f / t.sigma2[i]; }
    const float factor = (float)(1.0 / sf);
    float nDes = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; l++) { t.nfeat[l] = cvRoundF(nDes); sum += t.nfeat[l]; nDes *= factor; }
    t.nfeat[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
    const int HP = 15;
    int v, v0;
    const int vmax = (int)std::floor(HP * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(HP * std::sqrt(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) t.umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    return IVF_OK;
}

// ---- one batch context: geometry + device buffers for a fixed (params, image size, capacity) ----
struct Context {
    int device = 0, maxImg = 0, nSides = 1;
    bool introspection = false;
    bool needsCost = false;             // cell rows overshoot a level: only the introspection path (stale hY) is defined there
    Config hc{};
    Config* dc = nullptr;
    ResizeCoef* dTab = nullptr;           // packed cv::resize coefficients, all levels
    Buffers b{};
    uint8_t* dStage = nullptr;          // single-image host API staging (image + cost)
    uint8_t* hStage = nullptr;          // ... and its pinned host twin (+ room for the keypoints / descriptors coming back)
    size_t stageBytes = 0;
    static constexpr int kEvRing = 64;  // HIP event pairs around the FAST+NMS launch of the last kEvRing runs
    hipEvent_t evFast0[kEvRing] = {}, evFast1[kEvRing] = {};
    // r04: the blur (throughput-bound, needs only the pyramid) runs on a side stream beside the selection chain (quota, three tier
    // launches, level selection: latency-bound, ~285 us per 256 images) and joins in front of the descriptors.  The side stream is
    // LENT by the caller (the batched front end passes the internal stream of its next context): a stream of its own per context
    // cost configs[2] 15 % -- 10.4k -> 9.0k pairs/s just by existing, whether used or not: the runtime multiplexes streams onto a few
    // hardware queues, and three more streams put the FCN's stream behind front-end work
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    long long nRuns = 0;
    int markOwn = -1, markSide = -1;    // progress words of this context's stream / of the stream its blur is lent (experiment builds: IVF_MARK)
    hipStream_t lastStream = nullptr;

    int build(const Tables& t, int w, int h, int maxImages, int sides, int dev, bool withStereo, const int* variant = nullptr);
    int set_variant(const int* variant);
    void release();
    int run(const uint8_t* s0, const uint8_t* s1, const uint8_t* cost, size_t imageStride, int rowStride,
            size_t costStride, int costRowStride, int nImg, const uint8_t* hUseCost, hipStream_t st,
            hipEvent_t inputsConsumed = nullptr, hipStream_t sideStream = nullptr);
    // r06: per-side source override of the NEXT run (ivf_frontend_run_color): code 0 = grey with its own strides, 1 / 2 (| 4) = B,G,R / R,G,B interleaved
    // (k_ingest_color).  Consumed by run().
    struct SideSrc { const uint8_t* src = nullptr; size_t imageStride = 0; int rowStride = 0; int code = 0; } sideSrc[2];
    int check_status(int which = 0);
};

int Context::set_variant(const int* v)
{
    hc.varBlur = v[0] ? 1 : 0; hc.varRetain = v[1] ? 1 : 0; hc.varAtan = v[2] ? 1 : 0;
    if (dc) { HIPCHK(hipSetDevice(device)); HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(dc, &hc, sizeof(Config), hipMemcpyHostToDevice)); }
    return IVF_OK;
}

int Context::build(const Tables& t, int w, int h, int maxImages, int sides, int dev, bool withStereo, const int* variant)
{
    device = dev; maxImg = maxImages; nSides = sides;
    introspection = t.p.enable_introspection != 0;
    if (w < 1 || h < 1 || w > 4095 || h > 4095) return fail(IVF_E_INVALID, "image size %dx%d unsupported (max 4095)", w, h);
    Config& c = hc;
    memset(&c, 0, sizeof c);
    c.nlevels = t.p.nlevels; c.w = w; c.h = h; c.nfeatures = t.p.nfeatures;
    c.iniTh = t.p.ini_th_fast; c.minTh = t.p.min_th_fast; c.introspection = introspection ? 1 : 0;
    if (variant) { c.varBlur = variant[0] ? 1 : 0; c.varRetain = variant[1] ? 1 : 0; c.varAtan = variant[2] ? 1 : 0; }
    memcpy(c.umax, t.umax, sizeof c.umax);
    {   // k_describe carries umax as a packed constant (HALF_PATCH_SIZE = 15 is fixed): make sure it is this table
        const unsigned long long packed = 0x3689ABCDDEEEFFFFull;
        for (int i = 0; i < 16; i++)
            if ((int)((packed >> (4 * i)) & 15) != t.umax[i]) return fail(IVF_E_STATE, "umax table mismatch at %d", i);
    }
    const float imageRatio = (float)w / h;                       // ORBextractor.cc:884
    int off = 0, cellBase = 0, candBase = 0, kpBase = 0, tileBase = 0, btileBase = 0;
    for (int l = 0; l < c.nlevels; l++) {
        LevelGeom& G = c.lv[l];
        c.scale[l] = t.scale[l]; c.invScale[l] = t.invScale[l];
        G.scale = t.scale[l];
        G.w = cvRoundF((float)w * t.invScale[l]);                 // :1303
        G.h = cvRoundF((float)h * t.invScale[l]);
        if (G.w < 1 || G.h < 1) return fail(IVF_E_GEOMETRY, "pyramid level %d is empty for %dx%d", l, w, h);
        G.pitch = align_up(G.w, 64);
        G.off = off; off += G.pitch * G.h;
        G.nDesired = t.nfeat[l];
        G.kpBase = kpBase; kpBase += G.nDesired;
        G.scaledPatch = (int)(31 * t.scale[l]);                   // :1142
        G.cols = (int)std::sqrt((float)G.nDesired / (5 * imageRatio));   // :890
        G.rows = (int)(imageRatio * G.cols);                      // :891
        G.maxBX = G.w - kEdge; G.maxBY = G.h - kEdge;
        const int W = G.maxBX - kEdge, H = G.maxBY - kEdge;
        G.valid = (G.cols > 0 && G.rows > 0 && W > 0 && H > 0 && G.nDesired > 0) ? 1 : 0;
        G.btileBase = btileBase; G.btilesX = (G.w + kBlurTW - 1) / kBlurTW; G.btilesY = (G.h + kBlurTH - 1) / kBlurTH;
        btileBase += G.btilesX * G.btilesY;
        G.cellBase = cellBase; G.candBase = candBase; G.tileBase = tileBase;
        if (!G.valid) continue;
        G.cellW = (int)std::ceil((float)W / G.cols);              // :906-907
        G.cellH = (int)std::ceil((float)H / G.rows);
        G.nCells = G.rows * G.cols;
        G.cellWMagic = (unsigned)(0x100000000ull / (unsigned)G.cellW) + 1u;   // cellW, cellH >= 1 here; == 1 gives 0: see k_fast_nms
        G.cellHMagic = (unsigned)(0x100000000ull / (unsigned)G.cellH) + 1u;
        if (G.nCells > kMaxCells) return fail(IVF_E_INVALID, "level %d has %d cells (max %d)", l, G.nCells, kMaxCells);
        G.nfeaturesCell = (int)std::ceil((float)G.nDesired / G.nCells);   // :923
        // cell windows must stay inside [16, dim-16): otherwise the reference throws in rowRange/colRange
        // or reads outside the blurred clone (ORBextractor.cc:1033, 1276)
        // With introspection the stale-hY quirk gives every row the LAST row's window height, which is
        // negative or tiny exactly when the rows overshoot: the reference then runs (and finds nothing in those cells) as long
        // as a cost map comes with the image.  A per-call extractor with introspection on is therefore accepted and checked
        // per call; stereo contexts hold a non-introspective right side and are rejected here.
        const bool colsLeave = (G.cols - 1) * G.cellW > W, rowsLeave = (G.rows - 1) * G.cellH > H;
        const bool lastRowEmpty = H - (G.rows - 1) * G.cellH + 6 < 0;     // hY < 0 in the last row: stale negative height, rowRange throws
                                                                          // (hY == 0: empty windows, the level yields nothing -- k_quota)
        if (colsLeave || (rowsLeave && (!(introspection && sides == 1) || lastRowEmpty)))
            return fail(IVF_E_GEOMETRY, "level %d: %dx%d cells of %dx%d leave the %dx%d level", l, G.cols, G.rows,
                        G.cellW, G.cellH, G.w, G.h);
        if (rowsLeave) needsCost = true;
        G.domHLast = H - (G.rows - 1) * G.cellH;
        G.winHLast = G.domHLast + 6;
        G.domH[0] = G.cellH;
        G.domH[1] = G.domHLast;                                   // stale hY
        G.candCap = ((G.cellW + 1) / 2) * ((G.cellH + 1) / 2) + 1;
        c.maxCandCap = std::max(c.maxCandCap, G.candCap);
        cellBase += G.nCells; candBase += G.nCells * G.candCap;
        G.tilesX = (G.maxBX - 16 + kFastTW - 1) / kFastTW;
        G.tilesY = (G.maxBY - kEdge + kFastTH - 1) / kFastTH;
        tileBase += G.tilesX * G.tilesY;
    }
    c.pyrBytes = align_up(off + 64, 256);
    c.candTotal = std::max(candBase, 1);
    c.nCellsTotal = std::max(cellBase, 1);
    c.nTiles = tileBase;
    c.nBlurTiles = btileBase;
    for (int l = 0; l < kMaxLevels; l++) {
        c.cellBases[l] = (l < c.nlevels && c.lv[l].valid) ? c.lv[l].cellBase : INT_MAX;
        c.tileBases[l] = l < c.nlevels ? c.lv[l].tileBase : INT_MAX;
        c.btileBases[l] = l < c.nlevels ? c.lv[l].btileBase : INT_MAX;
    }

    // cv::resize coefficient table (OpenCV resize.cpp, INTER_LINEAR 8U): per output column
    // {clamped sx, a0, a1}, per output row {clipped sy, b0, b1}; x clamps f to 0 at the right edge, y does not.
    std::vector<ResizeCoef> tab;
    auto sat = [](float v) { int i = cvRoundF(v); return (unsigned)(unsigned short)std::min(32767, std::max(-32768, i)); };
    for (int l = 1; l < c.nlevels; l++) {
        LevelGeom& D = c.lv[l]; const LevelGeom& S = c.lv[l - 1];
        // cv::resize: inv_scale = (double)dsize / ssize, scale = 1. / inv_scale (not the direct quotient: the doubles can differ in the last bit)
        const double sx_ = 1. / ((double)D.w / S.w), sy_ = 1. / ((double)D.h / S.h);
        D.rtX = (int)tab.size();
        for (int dx = 0; dx < D.w; dx++) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int sx = cvFloorF(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
            tab.push_back((ResizeCoef)sx | ((ResizeCoef)sat((1.f - fx) * 2048.f) << 16) | ((ResizeCoef)sat(fx * 2048.f) << 32));
        }
        D.rtY = (int)tab.size();
        for (int dy = 0; dy < D.h; dy++) {
            float fy = (float)((dy + 0.5) * sy_ - 0.5);
            int sy = cvFloorF(fy);
            fy -= sy;
            // rows are clipped, coefficients are not (VResize reads clip(sy), clip(sy+1)); sy >= 0 for down-scaling
            const int y0 = std::min(std::max(sy, 0), S.h - 1);
            ResizeCoef b0 = sat((1.f - fy) * 2048.f), b1 = sat(fy * 2048.f);
            if (sy < 0) { /* both taps clip to row 0 */ }
            tab.push_back((ResizeCoef)y0 | (b0 << 16) | (b1 << 32));
        }
    }
    if (tab.empty()) tab.push_back(0);

    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc(&dc, sizeof(Config)));
    HIPCHK(hipMemcpy(dc, &hc, sizeof(Config), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&dTab, tab.size() * sizeof(ResizeCoef)));
    HIPCHK(hipMemcpy(dTab, tab.data(), tab.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice));
    const size_t nI = (size_t)maxImg, nf = (size_t)c.nfeatures, blob = (size_t)c.pyrBytes * nI;
    HIPCHK(hipMalloc(&b.pyr, blob + 256));          // + slack: k_stereo_match / k_describe read whole dwords that may end a few bytes past a plane's last row
    HIPCHK(hipMalloc(&b.blur, blob + 256));
    HIPCHK(hipMemset(b.pyr, 0, blob));
    if (introspection) { HIPCHK(hipMalloc(&b.qpyr, blob)); HIPCHK(hipMemset(b.qpyr, 0, blob)); }
    HIPCHK(hipMalloc(&b.tileList, nI * std::max(c.nTiles, 1) * (size_t)kTileCap * sizeof(unsigned)));
    HIPCHK(hipMalloc(&b.tileCnt, nI * std::max(c.nTiles, 1) * sizeof(int)));
    HIPCHK(hipMalloc(&b.cellCnt, nI * c.nCellsTotal * 2 * sizeof(int)));
    HIPCHK(hipMalloc(&b.cellInfo, nI * c.nCellsTotal * sizeof(int4)));
    HIPCHK(hipMalloc(&b.lvlTotal, nI * kMaxLevels * sizeof(int)));
    HIPCHK(hipMalloc(&b.lvl, nI * c.candTotal * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&b.slotPos, nI * nf * sizeof(unsigned)));
    HIPCHK(hipMalloc(&b.slotResp, nI * nf * sizeof(float)));
    HIPCHK(hipMalloc(&b.lvlCount, nI * kMaxLevels * sizeof(int)));
    HIPCHK(hipMemset(b.lvlCount, 0, nI * kMaxLevels * sizeof(int)));
    HIPCHK(hipMalloc(&b.useCost, (nI + 3) & ~(size_t)3));                // whole dwords: the kernels read the flags with scalar dword loads
    HIPCHK(hipMemset(b.useCost, 0, (nI + 3) & ~(size_t)3));
    HIPCHK(hipMalloc(&b.kps, nI * nf * sizeof(ivf_keypoint)));
    HIPCHK(hipMalloc(&b.desc, nI * nf * 32));
    HIPCHK(hipMalloc(&b.count, nI * sizeof(int)));
    HIPCHK(hipMemset(b.count, 0, nI * sizeof(int)));
    HIPCHK(hipMalloc(&b.quality, nI * nf * sizeof(float)));
    if (withStereo) {
        const size_t nP = std::max<size_t>(nI / 2, 1);
        HIPCHK(hipMalloc(&b.uright, nP * nf * sizeof(float)));
        HIPCHK(hipMalloc(&b.depth, nP * nf * sizeof(float)));
        HIPCHK(hipMalloc(&b.sad, nP * nf * sizeof(int)));
        HIPCHK(hipMalloc(&b.rowCnt, nP * (size_t)h * sizeof(int)));
        HIPCHK(hipMalloc(&b.rowList, nP * (size_t)h * kRowCap * sizeof(unsigned short)));
    }
    HIPCHK(hipMalloc(&b.status, 4 * sizeof(int)));
    HIPCHK(hipMemset(b.status, 0, 4 * sizeof(int)));
    b.hugeCount = b.status + 1;
    HIPCHK(hipMalloc(&b.tierList, 2 * nI * (size_t)c.nCellsTotal * sizeof(int)));
    if (c.maxCandCap > 4096) {          // kCellCapBig: cells of this geometry can outgrow the LDS selection paths
        HIPCHK(hipMalloc(&b.hugeList, (size_t)kHugeListCap * sizeof(int)));
        HIPCHK(hipMalloc(&b.hugeScratch, (size_t)kHugeSlots * 6 * c.maxCandCap * sizeof(unsigned)));
    }
    for (int i = 0; i < kEvRing; i++) { HIPCHK(hipEventCreate(&evFast0[i])); HIPCHK(hipEventCreate(&evFast1[i])); }
    {
        HIPCHK(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&evJoin, hipEventDisableTiming));
    }
    return IVF_OK;
}

void Context::release()
{
    (void)hipSetDevice(device);
    void* ptrs[] = {dc, dTab, b.pyr, b.qpyr, b.blur, b.tileList, b.tileCnt, b.cellCnt, b.cellInfo, b.lvlTotal, b.lvl, b.slotPos, b.slotResp, b.lvlCount,
                    b.useCost, b.kps, b.desc, b.count, b.quality, b.uright, b.depth, b.sad, b.rowCnt, b.rowList, b.status, b.hugeList, b.hugeScratch, b.tierList, dStage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (hStage) (void)hipHostFree(hStage);
    for (int i = 0; i < kEvRing; i++) {
        if (evFast0[i]) (void)hipEventDestroy(evFast0[i]);
        if (evFast1[i]) (void)hipEventDestroy(evFast1[i]);
    }
    if (evFork) (void)hipEventDestroy(evFork);
    if (evJoin) (void)hipEventDestroy(evJoin);
    *this = Context();
}

// Enqueue ORBextractor::operator() for nImg images (ORB/src/ORBextractor.cc:1224-1296)
int Context::run(const uint8_t* s0, const uint8_t* s1, const uint8_t* cost, size_t imageStride, int rowStride,
                 size_t costStride, int costRowStride, int nImg, const uint8_t* dUseCostSrc, hipStream_t st,
                 hipEvent_t inputsConsumed, hipStream_t sideStream)
{
    if (nImg < 1 || nImg > maxImg) return fail(IVF_E_INVALID, "batch of %d images outside [1,%d]", nImg, maxImg);
    HIPCHK(hipSetDevice(device));
    lastStream = st;
    const bool useQ = introspection && cost != nullptr;
    if (needsCost && !useQ)
        return fail(IVF_E_GEOMETRY, "cell rows leave a pyramid level: this geometry is defined only with a cost map (stale hY, ORBextractor.cc:935-999)");
    if (cost && !b.qpyr) {
        // a cost image with an extractor that ignores it (enableIntrospection = 0): mvKeyQualScore still reads it
        // (Frame.cc:130-143), so its level 0 is ingested.  One-time allocation, on the first such call.
        const size_t blob = (size_t)hc.pyrBytes * maxImg;
        HIPCHK(hipMalloc(&b.qpyr, blob));
        HIPCHK(hipMemsetAsync(b.qpyr, 0, blob, st));
    }
    if (cost) HIPCHK(hipMemcpyAsync(b.useCost, dUseCostSrc, nImg, hipMemcpyDeviceToDevice, st));
    else HIPCHK(hipMemsetAsync(b.useCost, 0, nImg, st));
    if (sideSrc[0].src || sideSrc[1].src) {
        for (int sd = 0; sd < nSides; sd++) {
            const SideSrc& q = sideSrc[sd];
            if (q.src && (q.code & 3)) launch_ingest_color(hc, dc, q.src, q.imageStride, q.rowStride, q.code, nImg, nSides, sd, b.pyr, st);
            else if (q.src) launch_ingest(hc, dc, b, q.src, q.src, q.imageStride, q.rowStride, nImg, nSides, b.pyr, st, 1 << sd);
            else launch_ingest(hc, dc, b, s0, s1, imageStride, rowStride, nImg, nSides, b.pyr, st, 1 << sd);
        }
        sideSrc[0] = SideSrc(); sideSrc[1] = SideSrc();
    } else
    launch_ingest(hc, dc, b, s0, s1, imageStride, rowStride, nImg, nSides, b.pyr, st);
    // r06: a cost map that already sits in the level-0 cost plane (ivf_frontend_cost_plane: the FCN wrote it there) needs no copy
    const bool costInPlace = cost && b.qpyr && cost == b.qpyr + hc.lv[0].off && costStride == (size_t)nSides * hc.pyrBytes && costRowStride == hc.lv[0].pitch;
    if (cost && !costInPlace) launch_ingest(hc, dc, b, cost, cost, costStride, costRowStride, nImg, nSides, b.qpyr, st);
    if (inputsConsumed) HIPCHK(hipEventRecord(inputsConsumed, st));   // caller buffers are free from here on
    IVF_MARK(st, markOwn, 1, nRuns);
    launch_pyramid(hc, dc, dTab, b.pyr, useQ ? b.qpyr : nullptr, b.useCost, nImg, st);   // + ComputeQualityImagePyramid :1325-1357
    IVF_MARK(st, markOwn, 2, nRuns);
    HIPCHK(hipMemsetAsync(b.cellCnt, 0, (size_t)nImg * hc.nCellsTotal * 2 * sizeof(int), st));
    HIPCHK(hipMemsetAsync(b.hugeCount, 0, 3 * sizeof(int), st));
    const int slot = (int)(nRuns % kEvRing);
    HIPCHK(hipEventRecord(evFast0[slot], st));
    launch_fast(hc, dc, b, nImg, st);
    HIPCHK(hipEventRecord(evFast1[slot], st));
    IVF_MARK(st, markOwn, 3, nRuns);
    const long long thisRun = nRuns;
    (void)thisRun;
    nRuns++;
    static const bool sideBlur = getenv("IVF_NO_SIDE_BLUR") == nullptr;
    hipStream_t side = sideStream;
    if (sideBlur && side && side != st && nImg > 2) {       // not for single frames: there the two event hand-overs cost more than the overlap gives (extraction 0.35 -> 0.63 ms)
        // fork: the blur only needs the pyramid (in order behind it on st); the next run's blur cannot overtake this run's descriptors,
        // because its fork event is recorded on st behind them
        HIPCHK(hipEventRecord(evFork, st));
        HIPCHK(hipStreamWaitEvent(side, evFork, 0));
        IVF_MARK(side, markSide, 1, thisRun);
        launch_blur(hc, dc, b, nImg, side, false);
        IVF_MARK(side, markSide, 2, thisRun);
        // whatever fails from here on, the owning stream joins the side stream's work before this call returns: an un-joined blur
        // would still be writing b.blur when the context is reused or released
        const hipError_t eRec = hipEventRecord(evJoin, side);
        if (eRec == hipSuccess) launch_select(hc, dc, b, nImg, st);
        IVF_MARK(st, markOwn, 4, thisRun);
        const hipError_t eWait = eRec == hipSuccess ? hipStreamWaitEvent(st, evJoin, 0) : eRec;
        if (eWait != hipSuccess) {
            (void)hipStreamSynchronize(side);
            return fail(IVF_E_NO_DEVICE, "joining the side-stream blur failed: %s", hipGetErrorString(eWait));
        }
        IVF_MARK(st, markOwn, 5, thisRun);
    } else {
        launch_select(hc, dc, b, nImg, st);
        IVF_MARK(st, markOwn, 4, thisRun);
        launch_blur(hc, dc, b, nImg, st, true);
        IVF_MARK(st, markOwn, 5, thisRun);
    }
    launch_describe(hc, dc, b, nullptr, 0, 0, nImg, nSides, st);
    IVF_MARK(st, markOwn, 6, thisRun);
    HIPCHK(hipGetLastError());
    return IVF_OK;
}

// Reads AND clears the device-side flags: an error is reported once, for the batch that raised it, and the handle stays
// usable (the flags used to be sticky: one bad image killed a streaming front end until it was re-created).
int Context::check_status(int which)
{
    int s = 0;
    HIPCHK(hipMemcpy(&s, b.status, sizeof(int), hipMemcpyDeviceToHost));
    if (!s) return IVF_OK;
    HIPCHK(hipMemset(b.status, 0, sizeof(int)));
    if (s & 4) return fail(IVF_E_CAPACITY, "batch context %d (run %lld): more than %d cells held over 4096 FAST survivors in one launch",
                           which, nRuns - 1, kHugeListCap);
    return fail(IVF_E_STATE, "batch context %d (run %lld): device-side consistency check failed (flags 0x%x)", which, nRuns - 1, s);
}

// Growable scratch per host thread: a device buffer (per-call entry points -- ivf_stereo_match, ivf_hamming_pairs, the matchers
// built on it, ivf_bow_transform, ivf_distinctive_descriptor -- upload their inputs here instead of paying a hipMalloc /
// hipFree set per call) and a pinned host buffer (row-wise unpacking of pitched device planes).
// Ordering assumption: every user works on the NULL stream and ends with a BLOCKING copy of its results, so two users on
// one thread can never overlap in a buffer.
// Ownership: the buffers live in slots of a process-wide pool.  A thread leases one slot on first use and its thread_local
// destructor only RETURNS the slot to the pool -- no HIP call at thread exit (for the main thread that would run during
// static destruction, possibly after the HIP runtime has gone) and no leak either: the reference starts two fresh
// std::threads per Frame (ORB/src/Frame.cc:116-124), and those now reuse the slots their predecessors gave back instead of
// pinning ~1 MB each for the life of the process.  The pool itself is never destroyed (freed with the process).
struct ScratchSlot { int device = -1; uint8_t* buf = nullptr; size_t cap = 0; uint8_t* pin = nullptr; size_t pinCap = 0; };
struct ScratchPool { std::mutex m; std::vector<ScratchSlot*> idle; int created = 0; };
ScratchPool& scratch_pool() { static ScratchPool* p = new ScratchPool; return *p; }
struct ScratchLease {
    ScratchSlot* s = nullptr;
    ~ScratchLease() { if (s) { ScratchPool& P = scratch_pool(); std::lock_guard<std::mutex> g(P.m); P.idle.push_back(s); s = nullptr; } }
};
ScratchSlot& my_scratch_slot()
{
    static thread_local ScratchLease lease;
    if (!lease.s) {
        ScratchPool& P = scratch_pool();
        std::lock_guard<std::mutex> g(P.m);
        if (!P.idle.empty()) { lease.s = P.idle.back(); P.idle.pop_back(); }
        else { lease.s = new ScratchSlot; P.created++; }
    }
    return *lease.s;
}

int thread_scratch(int device, size_t need, uint8_t** out)
{
    ScratchSlot& sc = my_scratch_slot();
    if (sc.device != device || sc.cap < need) {
        if (sc.buf) { (void)hipSetDevice(sc.device); (void)hipDeviceSynchronize(); (void)hipFree(sc.buf); sc.buf = nullptr; sc.cap = 0; }
        HIPCHK(hipSetDevice(device));
        const size_t cap = std::max(need + need / 2, (size_t)1 << 20);
        HIPCHK(hipMalloc(&sc.buf, cap));
        sc.cap = cap; sc.device = device;
    }
    *out = sc.buf;
    return IVF_OK;
}

int thread_pinned(size_t need, uint8_t** out)
{
    ScratchSlot& sc = my_scratch_slot();
    if (sc.pinCap < need) {
        if (sc.pin) { (void)hipHostFree(sc.pin); sc.pin = nullptr; sc.pinCap = 0; }
        const size_t cap = std::max(need + need / 2, (size_t)1 << 20);
        HIPCHK(hipHostMalloc((void**)&sc.pin, cap, hipHostMallocDefault));
        sc.pinCap = cap;
    }
    *out = sc.pin;
    return IVF_OK;
}

int have_device(int dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(IVF_E_NO_DEVICE, "no HIP device available (%s); libivfront has no CPU path",
                                               e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (dev < 0 || dev >= n) return fail(IVF_E_INVALID, "device_id %d outside [0,%d)", dev, n);
    return IVF_OK;
}

}  // namespace

static std::atomic<long long> g_launches{0};
void ivf::count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

int ivf::set_error(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

struct ivf_extractor {
    int var[3] = {0, 0, 0};              // OpenCV-version switches (blur, retain, atan)
    Tables t;
    int device;
    Context ctx;
    bool haveCtx = false;
    int w = 0, h = 0;
    bool lastHadCost = false, extracted = false;
    uint8_t* dOne = nullptr;            // device byte "1"
    // r04: every handle extracts on a stream of its own.  The reference runs the left and the right extractor on two host threads
    // (ORB/src/Frame.cc:116-124) so that they overlap; on the NULL stream the two launch sequences queued one behind the other
    // (bench.py latency_ms_batch1: 0.35 + 0.31 ms back to back).  ivf_extract returns only after its stream has drained, so every
    // later reader of the handle's buffers (ivf_stereo_match, pyramid copies: NULL stream) sees them complete.
    hipStream_t st = nullptr;
};

// kPipe batch contexts on kPipe internal streams: run k uses context k % kPipe, so the latency-bound tail of one batch
// (per-cell selection, descriptors, stereo) overlaps the throughput-bound head (ingest, pyramid, FAST, blur) of the
// next ones.  Results of a run stay valid until kPipe - 1 further runs have been enqueued.
#ifndef IVF_PIPE
#define IVF_PIPE 3
#endif
constexpr int kPipe = IVF_PIPE;         // batch contexts in flight
struct ivf_frontend {
    ivf_frontend_config cfg;
    Tables tl;
    Context ctx[kPipe];
    hipStream_t stream[kPipe] = {};
    hipEvent_t evIn[kPipe] = {}, evConsumed[kPipe] = {}, evDone[kPipe] = {};
    uint8_t* dFlags = nullptr;          // useCost flags when a cost batch is given: [L,R,L,R,...]
    int lastPairs = 0;
    int pairsOf[kPipe] = {};            // batch size of the run each context holds
    long long runs = 0;
    int last() const { return (int)((runs + kPipe - 1) % kPipe); }      // context of the most recent run
};

// ---- Frame grid (ORB/src/Frame.cc:415-430, 615-680; 64 x 48, Frame.h:43-44) ----
namespace {
constexpr int GC = 64, GR = 48;
struct Grid {
    std::vector<int> start, idx; float invW, invH;
    void build(const ivf_keypoint* k, int n, const ivf_bounds& bd)
    {
        invW = (float)GC / (bd.max_x - bd.min_x); invH = (float)GR / (bd.max_y - bd.min_y);
        start.assign(GC * GR + 1, 0); idx.assign(std::max(n, 1), 0);
        std::vector<int> cell(std::max(n, 1), -1);
        for (int i = 0; i < n; i++) {
            const int px = (int)roundf((k[i].x - bd.min_x) * invW), py = (int)roundf((k[i].y - bd.min_y) * invH);
            if (px < 0 || px >= GC || py < 0 || py >= GR) continue;
            cell[i] = px * GR + py; start[cell[i] + 1]++;
        }
        for (int c = 0; c < GC * GR; c++) start[c + 1] += start[c];
        std::vector<int> fill(GC * GR, 0);
        for (int i = 0; i < n; i++) if (cell[i] >= 0) idx[start[cell[i]] + fill[cell[i]]++] = i;
    }
    template <class F> void query(const ivf_keypoint* k, const ivf_bounds& bd, float x, float y, float r, int minL, int maxL, F f) const
    {
        const int x0 = std::max(0, (int)floorf((x - bd.min_x - r) * invW)); if (x0 >= GC) return;
        const int x1 = std::min(GC - 1, (int)ceilf((x - bd.min_x + r) * invW)); if (x1 < 0) return;
        const int y0 = std::max(0, (int)floorf((y - bd.min_y - r) * invH)); if (y0 >= GR) return;
        const int y1 = std::min(GR - 1, (int)ceilf((y - bd.min_y + r) * invH)); if (y1 < 0) return;
        const bool chk = (minL > 0) || (maxL >= 0);
        for (int ix = x0; ix <= x1; ix++)
            for (int iy = y0; iy <= y1; iy++) {
                const int c = ix * GR + iy;
                for (int j = start[c]; j < start[c + 1]; j++) {
                    const ivf_keypoint& kp = k[idx[j]];
                    if (chk) { if (kp.octave < minL) continue; if (maxL >= 0 && kp.octave > maxL) continue; }
                    if (fabsf(kp.x - x) < r && fabsf(kp.y - y) < r) f(idx[j]);
                }
            }
    }
};
}  // namespace

extern "C" {

int ivf_version(void) { return 100; }
long long ivf_debug_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
int ivf_debug_scratch_slots(void) { ScratchPool& P = scratch_pool(); std::lock_guard<std::mutex> g(P.m); return P.created; }
const char* ivf_last_error(void) { return g_err.c_str(); }
int ivf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ivf_extractor_create(const ivf_extractor_params* params, int device_id, ivf_extractor** out)
{
    if (!params || !out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    Tables t;
    int rc = make_tables(*params, t);
    if (rc) return rc;
    rc = have_device(device_id);
    if (rc) return rc;
    ivf_extractor* e = new ivf_extractor();
    e->t = t; e->device = device_id;
    *out = e;
    return IVF_OK;
}

void ivf_extractor_destroy(ivf_extractor* e)
{
    if (!e) return;
    if (e->haveCtx) e->ctx.release();
    if (e->dOne) (void)hipFree(e->dOne);
    if (e->st) (void)hipStreamDestroy(e->st);
    delete e;
}

int ivf_extractor_set_opencv_variant(ivf_extractor* e, int blur, int retain_best, int atan2)
{
    if (!e) return fail(IVF_E_INVALID, "null handle");
    e->var[0] = blur; e->var[1] = retain_best; e->var[2] = atan2;
    return e->haveCtx ? e->ctx.set_variant(e->var) : IVF_OK;
}

int ivf_frontend_set_opencv_variant(ivf_frontend* fe, int blur, int retain_best, int atan2)
{
    if (!fe) return fail(IVF_E_INVALID, "null handle");
    const int v[3] = {blur, retain_best, atan2};
    for (int k = 0; k < kPipe; k++) { const int rc = fe->ctx[k].set_variant(v); if (rc) return rc; }
    return IVF_OK;
}

int ivf_extractor_get_levels(const ivf_extractor* e) { return e ? e->t.p.nlevels : fail(IVF_E_INVALID, "null handle"); }
float ivf_extractor_get_scale_factor(const ivf_extractor* e) { return e ? e->t.p.scale_factor : 0.f; }

int ivf_extractor_get_scale_tables(const ivf_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2)
{
    if (!e) return fail(IVF_E_INVALID, "null handle");
    const int n = e->t.p.nlevels;
    if (scale) memcpy(scale, e->t.scale, n * sizeof(float));
    if (inv_scale) memcpy(inv_scale, e->t.invScale, n * sizeof(float));
    if (sigma2) memcpy(sigma2, e->t.sigma2, n * sizeof(float));
    if (inv_sigma2) memcpy(inv_sigma2, e->t.invSigma2, n * sizeof(float));
    return IVF_OK;
}

int ivf_extractor_get_feature_tables(const ivf_extractor* e, int32_t* features_per_level, int32_t* umax16)
{
    if (!e) return fail(IVF_E_INVALID, "null handle");
    if (features_per_level) memcpy(features_per_level, e->t.nfeat, e->t.p.nlevels * sizeof(int));
    if (umax16) memcpy(umax16, e->t.umax, 16 * sizeof(int));
    return IVF_OK;
}

int ivf_extract(ivf_extractor* e, const uint8_t* image, int width, int height, int stride,
                const uint8_t* cost, int cost_stride, ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!e || !n_out) return fail(IVF_E_INVALID, "null argument");
    *n_out = 0;
    if (!image || width <= 0 || height <= 0) return IVF_OK;      // empty image: silent return (:1227-1228)
    if (stride < width || (cost && cost_stride < width)) return fail(IVF_E_INVALID, "stride smaller than width");
    int rc = have_device(e->device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->device));
    if (!e->haveCtx || e->w != width || e->h != height) {
        if (e->haveCtx) { e->ctx.release(); e->haveCtx = false; }
        rc = e->ctx.build(e->t, width, height, 1, 1, e->device, true, e->var);
        if (rc) { e->ctx.release(); return rc; }
        // staging is part of the context: the handle only counts as built once all of it exists (release() frees what does)
        e->ctx.stageBytes = ((size_t)width * height * 2 + 15) & ~(size_t)15;       // keypoints / counters behind it stay 16-byte aligned
        if (hipMalloc(&e->ctx.dStage, e->ctx.stageBytes) != hipSuccess ||
            hipHostMalloc((void**)&e->ctx.hStage, e->ctx.stageBytes + (size_t)e->t.p.nfeatures * (sizeof(ivf_keypoint) + 32) + 64, hipHostMallocDefault) != hipSuccess ||
            (!e->dOne && (hipMalloc(&e->dOne, 1) != hipSuccess || hipMemset(e->dOne, 1, 1) != hipSuccess)) ||
            (!e->st && hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess)) {
            e->ctx.release();
            return fail(IVF_E_NO_DEVICE, "staging allocation failed for a %dx%d extractor", width, height);
        }
        e->haveCtx = true; e->w = width; e->h = height;
        HIPCHK(hipDeviceSynchronize());      // the context's buffers were cleared on the NULL stream: done before the handle's own stream touches them
    }
    e->extracted = false;
    Context& c = e->ctx;
    uint8_t* dImg = c.dStage;
    uint8_t* dCost = c.dStage + (size_t)width * height;
    // the caller's image is pageable memory (a cv::Mat): rows go through the handle's pinned staging buffer -- one host memcpy
    // and one DMA instead of a row-by-row pageable hipMemcpy2D (measured 2.9 ms of a 3.1 ms call); results come back the same way
    const size_t px = (size_t)width * height, nfe = (size_t)e->t.p.nfeatures;
    uint8_t* hImg = c.hStage; uint8_t* hCost = hImg + px;
    ivf_keypoint* hK = (ivf_keypoint*)(c.hStage + c.stageBytes); uint8_t* hD = (uint8_t*)(hK + nfe); int* hN = (int*)(hD + nfe * 32);
    for (int y = 0; y < height; y++) memcpy(hImg + (size_t)y * width, image + (size_t)y * stride, (size_t)width);
    const bool useCost = cost && e->t.p.enable_introspection;
    if (useCost) for (int y = 0; y < height; y++) memcpy(hCost + (size_t)y * width, cost + (size_t)y * cost_stride, (size_t)width);
    // from the first enqueue on, every way out of this call leaves the handle's stream DRAINED (r04 ADVICE): the next call memcpy's
    // into hStage and the NULL-stream readers (ivf_stereo_match, the pyramid copies) touch the context's buffers -- work of a call
    // that failed half-way must not still be in flight then (the stream is non-blocking: the NULL stream does not order it)
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{e->st};
    HIPCHK(hipMemcpyAsync(dImg, hImg, useCost ? 2 * px : px, hipMemcpyHostToDevice, e->st));
    rc = c.run(dImg, dImg, useCost ? dCost : nullptr, px, width, px, width, 1, e->dOne, e->st);
    if (rc) return rc;
    // counts are not known before the kernels finish: fetch the count and the full-capacity result arrays in one go (56 KB at N = 1000)
    HIPCHK(hipMemcpyAsync(hN, c.b.count, sizeof(int), hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(hN + 1, c.b.status, sizeof(int), hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(hK, c.b.kps, nfe * sizeof(ivf_keypoint), hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(hD, c.b.desc, nfe * 32, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    const int n = *hN;
    if (hN[1]) { rc = c.check_status(); if (rc) return rc; }      // device-side flags raised: read, clear and report them
    e->lastHadCost = useCost; e->extracted = true;
    if (n > cap) { *n_out = n; return fail(IVF_E_CAPACITY, "%d keypoints exceed caller capacity %d", n, cap); }
    if (n > 0) {
        if (!kps || !desc) return fail(IVF_E_INVALID, "null output buffers");
        memcpy(kps, hK, (size_t)n * sizeof(ivf_keypoint));
        memcpy(desc, hD, (size_t)n * 32);
    }
    *n_out = n;
    return IVF_OK;
}

static int copy_level(const ivf_extractor* e, const uint8_t* blob, int level, uint8_t* dst, int dst_stride, int* width, int* height)
{
    if (!e || !e->extracted || !blob) return fail(IVF_E_STATE, "no pyramid: call ivf_extract first");
    if (level < 0 || level >= e->t.p.nlevels) return fail(IVF_E_INVALID, "level %d out of range", level);
    const LevelGeom& G = e->ctx.hc.lv[level];
    if (width) *width = G.w;
    if (height) *height = G.h;
    if (!dst) return IVF_OK;
    if (dst_stride < G.w) return fail(IVF_E_INVALID, "dst_stride smaller than level width");
    HIPCHK(hipSetDevice(e->device));
    // pitched level -> caller rows through the calling thread's pinned scratch (a pageable 2-D copy runs row by row)
    uint8_t* pin = nullptr;
    const size_t bytes = (size_t)G.pitch * G.h;
    const int prc = thread_pinned(bytes, &pin);
    if (prc) return prc;
    HIPCHK(hipMemcpy(pin, blob + G.off, bytes, hipMemcpyDeviceToHost));
    for (int y = 0; y < G.h; y++) memcpy(dst + (size_t)y * dst_stride, pin + (size_t)y * G.pitch, (size_t)G.w);
    return IVF_OK;
}
int ivf_extractor_pyramid_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height)
{
    return copy_level(e, e ? e->ctx.b.pyr : nullptr, level, dst, dst_stride, width, height);
}
int ivf_extractor_quality_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height)
{
    if (e && !e->lastHadCost) return fail(IVF_E_STATE, "last extract had no cost map");
    return copy_level(e, e ? e->ctx.b.qpyr : nullptr, level, dst, dst_stride, width, height);
}
int ivf_extractor_blur_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height)
{
    if (e && e->extracted && level >= 0 && level < e->t.p.nlevels) {
        int cnt = 0;
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipMemcpy(&cnt, e->ctx.b.lvlCount + level, sizeof(int), hipMemcpyDeviceToHost));
        if (cnt == 0) return fail(IVF_E_STATE, "level %d had no keypoints in the last extract: not blurred (ORBextractor.cc:1267-1268)", level);
    }
    return copy_level(e, e ? e->ctx.b.blur : nullptr, level, dst, dst_stride, width, height);
}
int ivf_extractor_level_counts(const ivf_extractor* e, int32_t* counts)
{
    if (!e || !counts) return fail(IVF_E_INVALID, "null argument");
    if (!e->extracted) return fail(IVF_E_STATE, "call ivf_extract first");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(counts, e->ctx.b.lvlCount, e->t.p.nlevels * sizeof(int), hipMemcpyDeviceToHost));
    return IVF_OK;
}

int ivf_stereo_match(const ivf_extractor* left, const ivf_extractor* right,
                     const ivf_keypoint* kps_left, int n_left, const uint8_t* desc_left,
                     const ivf_keypoint* kps_right, int n_right, const uint8_t* desc_right,
                     float bf, float b, float* u_right, float* depth)
{
    if (!left || !right || !u_right || !depth || n_left < 0 || n_right < 0) return fail(IVF_E_INVALID, "bad argument");
    if (!left->extracted || !right->extracted) return fail(IVF_E_STATE, "both extractors must have run ivf_extract");
    if (left->device != right->device) return fail(IVF_E_INVALID, "extractors live on different devices");
    if (left->w != right->w || left->h != right->h || left->t.p.nlevels != right->t.p.nlevels)
        return fail(IVF_E_INVALID, "left/right geometry differs");
    const int nf = left->t.p.nfeatures;
    if (n_left > nf || n_right > 65534) return fail(IVF_E_CAPACITY, "too many keypoints");
    if (n_left == 0) return IVF_OK;
    HIPCHK(hipSetDevice(left->device));
    // caller-provided keypoints / descriptors are the matcher's inputs (mvKeys, mvDescriptors): upload them into the
    // calling thread's growable device scratch (no per-frame hipMalloc / hipFree, nothing to leak on an error path)
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nR = (size_t)std::max(n_right, 1);
    const size_t szKL = up((size_t)nf * sizeof(ivf_keypoint)), szKR = up(nR * sizeof(ivf_keypoint)), szDL = up((size_t)nf * 32),
                 szDR = up(nR * 32);
    uint8_t* base = nullptr;
    int rc = thread_scratch(left->device, szKL + szKR + szDL + szDR + 256, &base);
    if (rc) return rc;
    ivf_keypoint* dKL = (ivf_keypoint*)base; ivf_keypoint* dKR = (ivf_keypoint*)(base + szKL);
    uint8_t* dDL = base + szKL + szKR; uint8_t* dDR = dDL + szDL; int* dCnt = (int*)(dDR + szDR);
    HIPCHK(hipMemcpyAsync(dKL, kps_left, (size_t)n_left * sizeof(ivf_keypoint), hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(dDL, desc_left, (size_t)n_left * 32, hipMemcpyHostToDevice, nullptr));
    if (n_right) {
        HIPCHK(hipMemcpyAsync(dKR, kps_right, (size_t)n_right * sizeof(ivf_keypoint), hipMemcpyHostToDevice, nullptr));
        HIPCHK(hipMemcpyAsync(dDR, desc_right, (size_t)n_right * 32, hipMemcpyHostToDevice, nullptr));
    }
    const int cnt[2] = {n_left, n_right};
    HIPCHK(hipMemcpyAsync(dCnt, cnt, sizeof cnt, hipMemcpyHostToDevice, nullptr));
    const Context& cl = left->ctx;
    StereoArgs A;
    A.pyrL = cl.b.pyr; A.pyrR = right->ctx.b.pyr; A.pyrStride = 0;
    A.kpL = dKL; A.kpR = dKR; A.descL = dDL; A.descR = dDR; A.cntL = dCnt; A.cntR = dCnt + 1;
    A.kpStride = 0; A.cntStride = 0;
    A.uright = cl.b.uright; A.depth = cl.b.depth; A.sad = cl.b.sad; A.outStride = nf;
    A.bf = bf; A.bb = b; A.rowCnt = cl.b.rowCnt; A.rowList = cl.b.rowList;
    launch_stereo_args(cl.hc, cl.dc, A, 1, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(u_right, cl.b.uright, (size_t)n_left * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(depth, cl.b.depth, (size_t)n_left * sizeof(float), hipMemcpyDeviceToHost));
    return IVF_OK;
}

int ivf_hamming(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

int ivf_hamming_pairs(const uint8_t* desc_a, int n_a, const uint8_t* desc_b, int n_b,
                      const int32_t* pairs, int n_pairs, int32_t* dist, int device_id)
{
    if (n_pairs == 0) return IVF_OK;
    if (!desc_a || !desc_b || !pairs || !dist || n_a < 1 || n_b < 1 || n_pairs < 0) return fail(IVF_E_INVALID, "bad argument");
    for (int i = 0; i < n_pairs; i++)
        if (pairs[2 * i] < 0 || pairs[2 * i] >= n_a || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= n_b)
            return fail(IVF_E_INVALID, "pair %d indexes outside the descriptor arrays", i);
    int rc = have_device(device_id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_id));
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t szA = up((size_t)n_a * 32), szB = up((size_t)n_b * 32), szP = up((size_t)n_pairs * 2 * sizeof(int)),
                 szD = up((size_t)n_pairs * sizeof(int));
    uint8_t* scb = nullptr;
    rc = thread_scratch(device_id, szA + szB + szP + szD, &scb);
    if (rc) return rc;
    uint8_t* dA = scb; uint8_t* dB = dA + szA; int* dP = (int*)(dB + szB); int* dD = (int*)((uint8_t*)dP + szP);
    HIPCHK(hipMemcpyAsync(dA, desc_a, (size_t)n_a * 32, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(dB, desc_b, (size_t)n_b * 32, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(dP, pairs, (size_t)n_pairs * 2 * sizeof(int), hipMemcpyHostToDevice, nullptr));
    launch_hamming_pairs(dA, dB, dP, n_pairs, dD, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(dist, dD, (size_t)n_pairs * sizeof(int), hipMemcpyDeviceToHost));
    return IVF_OK;
}


int ivf_features_in_area(const ivf_keypoint* kps, int n, const ivf_bounds* bounds, float x, float y, float r,
                         int min_level, int max_level, int32_t* out, int cap, int* n_out)
{
    if (!kps || !bounds || !n_out || n < 0) return fail(IVF_E_INVALID, "bad argument");
    Grid g; g.build(kps, n, *bounds);
    int c = 0;
    g.query(kps, *bounds, x, y, r, min_level, max_level, [&](int i) { if (out && c < cap) out[c] = i; c++; });
    *n_out = c;
    return c > cap ? fail(IVF_E_CAPACITY, "%d indices exceed capacity %d", c, cap) : IVF_OK;
}

// order-dependent greedy assignment + rotation histogram of SearchByProjection(cur, last), replayed in query order
// (ORBmatcher.cc:1444-1511): candidates of query i = cand[qStart[i] .. qStart[i+1]) in GetFeaturesInArea order
static int replay_projection(const ivf_keypoint* cur_kps, const float* cur_uright, int n_q, const float* q_ur, const float* q_radius,
                             const float* q_angle, const uint8_t* q_blocks, int check_orientation, const std::vector<int>& qStart,
                             const std::vector<int>& cand, const std::vector<int>& dist, int32_t* cur_assign, uint8_t* removed = nullptr)
{
    const int HISTO_LENGTH = 30;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < n_q; i++) {
        int bestDist = 256, bestIdx2 = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++) {
            const int i2 = cand[p];
            if (cur_assign[i2] == -2) continue;
            if (cur_assign[i2] >= 0 && (!q_blocks || q_blocks[cur_assign[i2]])) continue;
            if (cur_uright[i2] > 0) { const float er = fabsf(q_ur[i] - cur_uright[i2]); if (er > q_radius[i]) continue; }
            if (dist[p] < bestDist) { bestDist = dist[p]; bestIdx2 = i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= 100) {
            cur_assign[bestIdx2] = i; nm++;
            if (check_orientation) {
                float rot = q_angle[i] - cur_kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {                                      // ComputeThreeMaxima :1654-1695
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { cur_assign[j] = -1; nm--; if (removed) removed[j] = 1; }
    }
    return nm;
}

// best / second best + ratio test of SearchByProjection(F, mapPoints), greedy in map-point order (ORBmatcher.cc:86-126):
// candidates of map point i = cand[qStart[i] .. qStart[i+1]) in GetFeaturesInArea order
static int replay_map_points(const ivf_keypoint* cur_kps, const float* cur_uright, int n_q, const float* q_ur, const float* q_radius,
                             const uint8_t* q_blocks, float nn_ratio, const std::vector<int>& qStart, const std::vector<int>& cand,
                             const std::vector<int>& dist, int32_t* cur_assign)
{
    int nm = 0;
    for (int i = 0; i < n_q; i++) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++) {
            const int idx = cand[p];
            if (cur_assign[idx] == -2) continue;
            if (cur_assign[idx] >= 0 && (!q_blocks || q_blocks[cur_assign[idx]])) continue;
            if (cur_uright[idx] > 0) { const float er = fabsf(q_ur[i] - cur_uright[idx]); if (er > q_radius[i]) continue; }
            const int d = dist[p];
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = cur_kps[idx].octave; bestIdx = idx; }
            else if (d < bestDist2) { bestLevel2 = cur_kps[idx].octave; bestDist2 = d; }
        }
        if (bestIdx >= 0 && bestDist <= 100) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            cur_assign[bestIdx] = i; nm++;
        }
    }
    return nm;
}

int ivf_search_by_projection(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                             const ivf_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                             const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                             const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                             int check_orientation, int32_t* cur_assign, int* nmatches, int device_id)
{
    return ivf_search_by_projection_ex(cur_kps, cur_desc, cur_uright, n_cur, bounds, n_q, q_u, q_v, q_ur, q_radius, q_min_level,
                                       q_max_level, q_angle, q_desc, q_valid, q_blocks, check_orientation, cur_assign, nullptr,
                                       nmatches, device_id);
}

int ivf_search_by_projection_ex(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                                const ivf_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                                const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                                const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                                int check_orientation, int32_t* cur_assign, uint8_t* cur_removed, int* nmatches, int device_id)
{
    if (cur_removed && n_cur > 0) memset(cur_removed, 0, (size_t)n_cur);
    if (!cur_kps || !cur_desc || !cur_uright || !bounds || !cur_assign || !nmatches || n_cur < 0 || n_q < 0)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    if (n_q == 0 || n_cur == 0) return IVF_OK;
    if (!q_u || !q_v || !q_ur || !q_radius || !q_min_level || !q_max_level || !q_angle || !q_desc)
        return fail(IVF_E_INVALID, "null query array");
    // 1. candidate windows in the reference's GetFeaturesInArea order (:1429-1437)
    Grid g; g.build(cur_kps, n_cur, *bounds);
    std::vector<int> qStart(n_q + 1, 0), pairs;
    for (int i = 0; i < n_q; i++) {
        qStart[i] = (int)pairs.size() / 2;
        if (q_valid && !q_valid[i]) continue;
        g.query(cur_kps, *bounds, q_u[i], q_v[i], q_radius[i], q_min_level[i], q_max_level[i],
                [&](int i2) { pairs.push_back(i); pairs.push_back(i2); });
    }
    qStart[n_q] = (int)pairs.size() / 2;
    const int nPairs = qStart[n_q];
    // 2. every window distance on the device (DescriptorDistance :1459-1461)
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(q_desc, n_q, cur_desc, n_cur, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    // 3. order-dependent greedy assignment + rotation histogram, replayed in query order (:1444-1511)
    std::vector<int> cand(std::max(nPairs, 1));
    for (int p = 0; p < nPairs; p++) cand[p] = pairs[2 * p + 1];
    const int nm = replay_projection(cur_kps, cur_uright, n_q, q_ur, q_radius, q_angle, q_blocks, check_orientation, qStart, cand, dist, cur_assign,
                                     cur_removed);
    *nmatches = nm;
    return IVF_OK;
}

// ---- device-resident frame: keypoints, descriptors and the 64x48 grid stay in HBM between searches -----------------
// A frame's device memory is ONE arena (keypoints | descriptors | uRight | grid start | grid index) with a stream of its own,
// and its query scratch a second one; both come from a per-process pool and go back to it in ivf_frame_destroy, so a tracker
// that makes a frame per image pays hipMalloc / hipStreamCreate only until the pool is warm (r02: five hipMalloc + a stream
// per frame = 0.5 ms).
namespace {
struct Arena { int device = -1; uint8_t* base = nullptr; size_t cap = 0; hipStream_t stream = nullptr; };
struct ArenaPool {
    std::mutex m; std::vector<Arena> idle;
    int acquire(int device, size_t need, Arena& out)
    {
        {
            std::lock_guard<std::mutex> g(m);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); i++)
                if (idle[i].device == device && idle[i].cap >= need && (best < 0 || idle[i].cap < idle[best].cap)) best = i;
            if (best >= 0) { out = idle[best]; idle.erase(idle.begin() + best); return IVF_OK; }
        }
        Arena a; a.device = device;
        a.cap = ((need + need / 4) + 65535) & ~(size_t)65535;         // slack: frames of slightly different sizes share arenas
        if (hipMalloc(&a.base, a.cap) != hipSuccess) return fail(IVF_E_NO_DEVICE, "hipMalloc of a %zu-byte frame arena failed", a.cap);
        if (hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipFree(a.base); return fail(IVF_E_NO_DEVICE, "stream creation failed"); }
        out = a;
        return IVF_OK;
    }
    void release(Arena& a)
    {
        if (!a.base) return;
        (void)hipStreamSynchronize(a.stream);                         // nothing of the old owner may still be in flight
        {
            std::lock_guard<std::mutex> g(m);
            if (idle.size() < 64) { idle.push_back(a); a = Arena(); return; }
        }
        (void)hipFree(a.base); (void)hipStreamDestroy(a.stream);
        a = Arena();
    }
};
ArenaPool* frame_pool_ptr() { static ArenaPool* p = new ArenaPool(); return p; }   // never destroyed: no HIP calls during static destruction
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

struct ivf_frame {
    int device = 0, n = 0;
    ivf_bounds bd{};
    float invW = 0, invH = 0;
    std::vector<ivf_keypoint> kps;      // host copies for the greedy replay (angle, uRight)
    std::vector<float> uright;
    std::vector<uint8_t> desc;          // host copy for the overflow fallback
    Arena mem, qmem;                    // frame data / query scratch (pooled)
    ivf_keypoint* dKps = nullptr; uint8_t* dDesc = nullptr; int *dStart = nullptr, *dIdx = nullptr;
    // query scratch, grown on demand
    int qCap = 0, cCap = 0;
    float *dQu = nullptr, *dQv = nullptr, *dQr = nullptr; int *dQmin = nullptr, *dQmax = nullptr; uint8_t *dQdesc = nullptr, *dQvalid = nullptr;
    int *dCount = nullptr, *dCand = nullptr;
    hipStream_t stream = nullptr;
    float* dUright = nullptr;           // frames made from a front-end batch: uRight stays on the device until a replay needs it
    bool hostKps = true, hostDesc = true;   // host mirrors present (false: fetched on first use, see frame_host)
};

// carve the frame arena for n keypoints
static int frame_alloc(ivf_frame* f, int n)
{
    const size_t nn = (size_t)std::max(n, 1);
    const size_t oK = 0, oD = oK + up256(nn * sizeof(ivf_keypoint)), oU = oD + up256(nn * 32), oS = oU + up256(nn * sizeof(float)),
                 oI = oS + up256((GC * GR + 1) * sizeof(int)), total = oI + up256(nn * sizeof(int));
    const int rc = frame_pool_ptr()->acquire(f->device, total, f->mem);
    if (rc) return rc;
    uint8_t* b = f->mem.base;
    f->dKps = (ivf_keypoint*)(b + oK); f->dDesc = b + oD; f->dUright = (float*)(b + oU); f->dStart = (int*)(b + oS); f->dIdx = (int*)(b + oI);
    f->stream = f->mem.stream;
    return IVF_OK;
}

// the greedy replays read angle / octave / uRight of the frame's keypoints on the host, the overflow fallback its descriptors:
// frames created from host arrays carry them; frames created from a front-end batch fetch them on first use (28 B per keypoint)
static int frame_host(ivf_frame* f, bool needDesc)
{
    if (!f->hostKps) {
        HIPCHK(hipSetDevice(f->device));
        f->kps.resize(std::max(f->n, 1)); f->uright.resize(std::max(f->n, 1));
        if (f->n > 0) {
            HIPCHK(hipMemcpyAsync(f->kps.data(), f->dKps, (size_t)f->n * sizeof(ivf_keypoint), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipMemcpyAsync(f->uright.data(), f->dUright, (size_t)f->n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipStreamSynchronize(f->stream));
        }
        f->hostKps = true;
    }
    if (needDesc && !f->hostDesc) {
        HIPCHK(hipSetDevice(f->device));
        f->desc.resize((size_t)std::max(f->n, 1) * 32);
        if (f->n > 0) HIPCHK(hipMemcpy(f->desc.data(), f->dDesc, (size_t)f->n * 32, hipMemcpyDeviceToHost));
        f->hostDesc = true;
    }
    return IVF_OK;
}

void ivf_frame_destroy(ivf_frame* f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    frame_pool_ptr()->release(f->mem);
    frame_pool_ptr()->release(f->qmem);
    delete f;
}

int ivf_frame_count(const ivf_frame* f)
{
    return f ? f->n : fail(IVF_E_INVALID, "null handle");
}

int ivf_frame_create(const ivf_keypoint* kps, const uint8_t* desc, const float* uright, int n, const ivf_bounds* bounds,
                     int device_id, ivf_frame** out)
{
    if (!out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (!bounds || n < 0 || (n > 0 && (!kps || !desc || !uright))) return fail(IVF_E_INVALID, "bad argument");
    if (!(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y)) return fail(IVF_E_INVALID, "empty image bounds");
    int rc = have_device(device_id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_id));
    ivf_frame* f = new ivf_frame();
    f->device = device_id; f->n = n; f->bd = *bounds;
    f->invW = (float)GC / (bounds->max_x - bounds->min_x); f->invH = (float)GR / (bounds->max_y - bounds->min_y);   // Frame.cc:208-209
    f->kps.assign(kps, kps + n); f->uright.assign(uright, uright + n); f->desc.assign(desc, desc + (size_t)n * 32);
    rc = frame_alloc(f, n);
    if (rc) { ivf_frame_destroy(f); return rc; }
    if (n > 0) {
        if (hipMemcpyAsync(f->dKps, kps, (size_t)n * sizeof(ivf_keypoint), hipMemcpyHostToDevice, f->stream) != hipSuccess ||
            hipMemcpyAsync(f->dDesc, desc, (size_t)n * 32, hipMemcpyHostToDevice, f->stream) != hipSuccess) {
            ivf_frame_destroy(f);
            return fail(IVF_E_NO_DEVICE, "frame upload failed");
        }
    }
    launch_grid_build(f->dKps, n, bounds->min_x, bounds->min_y, f->invW, f->invH, f->dStart, f->dIdx, f->stream);   // AssignFeaturesToGrid
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(f->stream) != hipSuccess) {
        ivf_frame_destroy(f);
        return fail(IVF_E_NO_DEVICE, "grid build failed");
    }
    *out = f;
    return IVF_OK;
}

int ivf_frame_grid(const ivf_frame* f, int32_t* cell_start, int32_t* cell_index)
{
    if (!f || !cell_start || !cell_index) return fail(IVF_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(f->device));
    HIPCHK(hipMemcpy(cell_start, f->dStart, (GC * GR + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (f->n > 0) HIPCHK(hipMemcpy(cell_index, f->dIdx, (size_t)f->n * sizeof(int), hipMemcpyDeviceToHost));
    return IVF_OK;
}

// windows (GetFeaturesInArea) and distances (DescriptorDistance) of every query against the resident frame, on the device:
// candidates of query i = cand / dist [qStart[i] .. qStart[i+1]) in the reference's order
static int frame_candidates(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                            const int32_t* q_min_level, const int32_t* q_max_level, const uint8_t* q_desc, const uint8_t* q_valid,
                            std::vector<int>& qStart, std::vector<int>& cand, std::vector<int>& dist)
{
    HIPCHK(hipSetDevice(f->device));
    static const int capEnv = getenv("IVF_FRAME_WINDOW_CAP") ? atoi(getenv("IVF_FRAME_WINDOW_CAP")) : 0;   // tests: force the overflow path
    const int cap = capEnv > 0 ? capEnv : 128;
    if (n_q > f->qCap || cap > f->cCap) {
        frame_pool_ptr()->release(f->qmem);
        f->qCap = 0;
        const size_t nq = (size_t)n_q + 256;
        const size_t o1 = up256(nq * 4), oDesc = 5 * o1, oValid = oDesc + up256(nq * 32), oCount = oValid + up256(nq), oCand = oCount + o1,
                     total = oCand + up256(nq * cap * 8);
        const int prc = frame_pool_ptr()->acquire(f->device, total, f->qmem);
        if (prc) return prc;
        uint8_t* b = f->qmem.base;
        f->dQu = (float*)b; f->dQv = (float*)(b + o1); f->dQr = (float*)(b + 2 * o1); f->dQmin = (int*)(b + 3 * o1); f->dQmax = (int*)(b + 4 * o1);
        f->dQdesc = b + oDesc; f->dQvalid = b + oValid; f->dCount = (int*)(b + oCount); f->dCand = (int*)(b + oCand);
        f->qCap = (int)nq; f->cCap = cap;
    }
    hipStream_t st = f->stream;
    const size_t nq = (size_t)n_q;
    HIPCHK(hipMemcpyAsync(f->dQu, q_u, nq * 4, hipMemcpyHostToDevice, st)); HIPCHK(hipMemcpyAsync(f->dQv, q_v, nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->dQr, q_radius, nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->dQmin, q_min_level, nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->dQmax, q_max_level, nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->dQdesc, q_desc, nq * 32, hipMemcpyHostToDevice, st));
    if (q_valid) HIPCHK(hipMemcpyAsync(f->dQvalid, q_valid, nq, hipMemcpyHostToDevice, st));
    launch_grid_window(f->dKps, f->dDesc, f->dStart, f->dIdx, f->bd.min_x, f->bd.min_y, f->invW, f->invH, n_q, f->dQu, f->dQv, f->dQr,
                       f->dQmin, f->dQmax, f->dQdesc, q_valid ? f->dQvalid : nullptr, f->cCap, f->dCount, f->dCand, st);
    HIPCHK(hipGetLastError());
    std::vector<int> count(n_q), raw(nq * f->cCap * 2);
    HIPCHK(hipMemcpyAsync(count.data(), f->dCount, nq * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(raw.data(), f->dCand, nq * f->cCap * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    qStart.assign(n_q + 1, 0); cand.clear(); dist.clear();
    Grid g; bool haveGrid = false;
    for (int i = 0; i < n_q; i++) {
        qStart[i] = (int)cand.size();
        if (count[i] <= f->cCap) {
            for (int k = 0; k < count[i]; k++) { cand.push_back(raw[((size_t)i * f->cCap + k) * 2]); dist.push_back(raw[((size_t)i * f->cCap + k) * 2 + 1]); }
        } else {
            // a window with more candidates than the device list holds: this query again through the host grid
            if (!haveGrid) { const int hrc = frame_host(f, true); if (hrc) return hrc; g.build(f->kps.data(), f->n, f->bd); haveGrid = true; }
            g.query(f->kps.data(), f->bd, q_u[i], q_v[i], q_radius[i], q_min_level[i], q_max_level[i], [&](int i2) {
                cand.push_back(i2); dist.push_back(ivf_hamming(q_desc + (size_t)i * 32, f->desc.data() + (size_t)i2 * 32)); });
        }
    }
    qStart[n_q] = (int)cand.size();
    if (cand.empty()) { cand.push_back(0); dist.push_back(0); }
    return IVF_OK;
}

// Where the window candidates of a set of queries (Frame / KeyFrame::GetFeaturesInArea order, octave range applied) and their
// Hamming distances come from: a frame handed over as host arrays (host grid + k_hamming_pairs), or a device-resident
// ivf_frame (k_grid_window: windows AND distances on the device, nothing but the queries is uploaded).
namespace {
struct CandSource {
    const ivf_keypoint* kps = nullptr; const uint8_t* desc = nullptr; const float* uright = nullptr; int n = 0;
    const ivf_bounds* bd = nullptr; int device = 0; ivf_frame* frame = nullptr;
    static CandSource host(const ivf_keypoint* k, const uint8_t* d, const float* ur, int n, const ivf_bounds* b, int dev)
    { CandSource s; s.kps = k; s.desc = d; s.uright = ur; s.n = n; s.bd = b; s.device = dev; return s; }
    static int resident(ivf_frame* f, CandSource& s)
    {
        const int rc = frame_host(f, false); if (rc) return rc;
        s.kps = f->kps.data(); s.uright = f->uright.data(); s.n = f->n; s.bd = &f->bd; s.device = f->device; s.frame = f;
        return IVF_OK;
    }
    // candidates of query i = cand / dist [qStart[i] .. qStart[i+1]); lo / hi = GetFeaturesInArea's minLevel / maxLevel per query
    int get(int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* lo, const int32_t* hi,
            const uint8_t* q_desc, const uint8_t* q_valid, std::vector<int>& qStart, std::vector<int>& cand, std::vector<int>& dist) const
    {
        if (frame) return frame_candidates(frame, n_q, q_u, q_v, q_radius, lo, hi, q_desc, q_valid, qStart, cand, dist);
        Grid g; g.build(kps, n, *bd);
        std::vector<int> pairs;
        qStart.assign(n_q + 1, 0);
        for (int i = 0; i < n_q; i++) {
            qStart[i] = (int)pairs.size() / 2;
            if (q_valid && !q_valid[i]) continue;
            g.query(kps, *bd, q_u[i], q_v[i], q_radius[i], lo[i], hi[i], [&](int i2) { pairs.push_back(i); pairs.push_back(i2); });
        }
        qStart[n_q] = (int)pairs.size() / 2;
        const int nPairs = qStart[n_q];
        dist.assign(std::max(nPairs, 1), 0); cand.assign(std::max(nPairs, 1), 0);
        const int rc = ivf_hamming_pairs(q_desc, n_q, desc, n, pairs.data(), nPairs, dist.data(), device);
        if (rc) return rc;
        for (int p = 0; p < nPairs; p++) cand[p] = pairs[2 * p + 1];
        return IVF_OK;
    }
};
inline void level_window(int n_q, const int32_t* q_level, int below, int above, std::vector<int32_t>& lo, std::vector<int32_t>& hi)
{
    lo.resize(std::max(n_q, 1)); hi.resize(std::max(n_q, 1));
    for (int i = 0; i < n_q; i++) { lo[i] = q_level[i] - below; hi[i] = q_level[i] + above; }
}
}  // namespace

int ivf_frame_search_by_projection(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                                   const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                                   const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                                   int check_orientation, int32_t* cur_assign, int* nmatches)
{
    if (!f || !cur_assign || !nmatches || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    if (n_q == 0 || f->n == 0) return IVF_OK;
    if (!q_u || !q_v || !q_ur || !q_radius || !q_min_level || !q_max_level || !q_angle || !q_desc)
        return fail(IVF_E_INVALID, "null query array");
    std::vector<int> qStart, cand, dist;
    int rc = frame_host(f, false);
    if (rc) return rc;
    rc = frame_candidates(f, n_q, q_u, q_v, q_radius, q_min_level, q_max_level, q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    // greedy assignment + rotation histogram on the host (:1444-1511)
    *nmatches = replay_projection(f->kps.data(), f->uright.data(), n_q, q_ur, q_radius, q_angle, q_blocks, check_orientation, qStart,
                                  cand, dist, cur_assign);
    return IVF_OK;
}

int ivf_frame_search_map_points(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                                const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                                float nn_ratio, int32_t* cur_assign, int* nmatches)
{
    if (!f || !cur_assign || !nmatches || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    if (n_q == 0 || f->n == 0) return IVF_OK;
    if (!q_u || !q_v || !q_ur || !q_radius || !q_level || !q_desc) return fail(IVF_E_INVALID, "null query array");
    std::vector<int32_t> lo(n_q), hi(n_q);
    for (int i = 0; i < n_q; i++) { lo[i] = q_level[i] - 1; hi[i] = q_level[i]; }       // levels [pred - 1, pred] (:72-73)
    std::vector<int> qStart, cand, dist;
    int rc = frame_host(f, false);
    if (rc) return rc;
    rc = frame_candidates(f, n_q, q_u, q_v, q_radius, lo.data(), hi.data(), q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    *nmatches = replay_map_points(f->kps.data(), f->uright.data(), n_q, q_ur, q_radius, q_blocks, nn_ratio, qStart, cand, dist, cur_assign);
    return IVF_OK;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (ORB/src/ORBmatcher.cc:45-135)
int ivf_search_map_points(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                          const ivf_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                          const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                          const uint8_t* q_valid, const uint8_t* q_blocks, float nn_ratio,
                          int32_t* cur_assign, int* nmatches, int device_id)
{
    if (!cur_kps || !cur_desc || !cur_uright || !bounds || !cur_assign || !nmatches || n_cur < 0 || n_q < 0)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    if (n_q == 0 || n_cur == 0) return IVF_OK;
    if (!q_u || !q_v || !q_ur || !q_radius || !q_level || !q_desc) return fail(IVF_E_INVALID, "null query array");
    // 1. candidate windows in GetFeaturesInArea order, levels [pred-1, pred] (:72-73)
    Grid g; g.build(cur_kps, n_cur, *bounds);
    std::vector<int> qStart(n_q + 1, 0), pairs;
    for (int i = 0; i < n_q; i++) {
        qStart[i] = (int)pairs.size() / 2;
        if (q_valid && !q_valid[i]) continue;
        g.query(cur_kps, *bounds, q_u[i], q_v[i], q_radius[i], q_level[i] - 1, q_level[i],
                [&](int i2) { pairs.push_back(i); pairs.push_back(i2); });
    }
    qStart[n_q] = (int)pairs.size() / 2;
    const int nPairs = qStart[n_q];
    // 2. all window distances on the device
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(q_desc, n_q, cur_desc, n_cur, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    // 3. best / second best + ratio test, greedy in map-point order (:86-126)
    std::vector<int> cand(std::max(nPairs, 1));
    for (int p = 0; p < nPairs; p++) cand[p] = pairs[2 * p + 1];
    const int nm = replay_map_points(cur_kps, cur_uright, n_q, q_ur, q_radius, q_blocks, nn_ratio, qStart, cand, dist, cur_assign);
    *nmatches = nm;
    return IVF_OK;
}

// ORBmatcher::SearchForInitialization (ORB/src/ORBmatcher.cc:410-519)
int ivf_search_for_initialization(const ivf_keypoint* kps1, const uint8_t* desc1, int n1,
                                  const ivf_keypoint* kps2, const uint8_t* desc2, int n2, const ivf_bounds* bounds2,
                                  float* prev_matched_xy, int window_size, float nn_ratio, int check_orientation,
                                  int32_t* matches12, int* nmatches, int device_id)
{
    if (!kps1 || !desc1 || !kps2 || !desc2 || !bounds2 || !prev_matched_xy || !matches12 || !nmatches || n1 < 0 || n2 < 0)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return IVF_OK;
    // 1. windows of the octave-0 keypoints of F1 in F2's grid, GetFeaturesInArea order (:426-430)
    Grid g; g.build(kps2, n2, *bounds2);
    std::vector<int> qStart(n1 + 1, 0), pairs;
    for (int i1 = 0; i1 < n1; i1++) {
        qStart[i1] = (int)pairs.size() / 2;
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        g.query(kps2, *bounds2, prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window_size, level1, level1,
                [&](int i2) { pairs.push_back(i1); pairs.push_back(i2); });
    }
    qStart[n1] = (int)pairs.size() / 2;
    const int nPairs = qStart[n1];
    // 2. every window distance on the device (:445)
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(desc1, n1, desc2, n2, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    // 3. order-dependent replay: best / second best against the distances already claimed, stealing, histogram (:437-510)
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> matchedDist(n2, INT_MAX), matches21(n2, -1);
    int nm = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (qStart[i1] == qStart[i1 + 1]) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int p = qStart[i1]; p < qStart[i1 + 1]; p++) {
            const int i2 = pairs[2 * p + 1], d = dist[p];
            if (matchedDist[i2] <= d) continue;
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nn_ratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nm--; }
            matches12[i1] = bestIdx2; matches21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist; nm++;
            if (check_orientation) {
                float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);
            }
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx1 : rotHist[i])
                    if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prev_matched_xy[2 * i1] = kps2[matches12[i1]].x; prev_matched_xy[2 * i1 + 1] = kps2[matches12[i1]].y; }
    *nmatches = nm;
    return IVF_OK;
}

// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORB/src/ORBmatcher.cc:296-404), flat queries
static int keyframe_points_impl(const CandSource& S, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                                const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches)
{
    *nmatches = 0;
    if (n_q == 0 || S.n == 0) return IVF_OK;
    if (!q_u || !q_v || !q_radius || !q_level || !q_desc) return fail(IVF_E_INVALID, "null query array");
    // 1. + 2. windows (KeyFrame::GetFeaturesInArea, no level arguments) filtered by octave in [level-1, level] (:384-385), distances
    std::vector<int32_t> lo, hi; level_window(n_q, q_level, 1, 0, lo, hi);
    std::vector<int> qStart, cand, dist;
    const int rc = S.get(n_q, q_u, q_v, q_radius, lo.data(), hi.data(), q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    // 3. greedy replay in candidate order: occupied keypoints are skipped (:379-380)
    int nm = 0;
    for (int i = 0; i < n_q; i++) {
        int bestDist = 256, bestIdx = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++) {
            const int idx = cand[p];
            if (matched[idx] != -1) continue;
            if (dist[p] < bestDist) { bestDist = dist[p]; bestIdx = idx; }
        }
        if (bestDist <= 50) { matched[bestIdx] = i; nm++; }
    }
    *nmatches = nm;
    return IVF_OK;
}
int ivf_search_keyframe_points(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, int n_kf, const ivf_bounds* bounds,
                               int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                               const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches, int device_id)
{
    if (!kf_kps || !kf_desc || !bounds || !matched || !nmatches || n_kf < 0 || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    return keyframe_points_impl(CandSource::host(kf_kps, kf_desc, nullptr, n_kf, bounds, device_id), n_q, q_u, q_v, q_radius, q_level,
                                q_desc, q_valid, matched, nmatches);
}
int ivf_frame_search_keyframe_points(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                                     const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches)
{
    if (!f || !matched || !nmatches || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    CandSource S; const int rc = CandSource::resident(f, S); if (rc) return rc;
    return keyframe_points_impl(S, n_q, q_u, q_v, q_radius, q_level, q_desc, q_valid, matched, nmatches);
}

// ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) matching core (ORB/src/ORBmatcher.cc:893-955), flat queries
static int fuse_impl(const CandSource& S, const float* inv_level_sigma2, int n_levels, int n_q, const float* q_u, const float* q_v,
                     const float* q_ur, const float* q_radius, const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid,
                     int32_t* best_idx, int32_t* best_dist)
{
    const bool gate = inv_level_sigma2 != nullptr;               // NULL: Fuse(KF, Scw, ...) (:983-1106) has no chi-square gate
    if (gate && (!S.uright || !q_ur || n_levels < 1)) return fail(IVF_E_INVALID, "the chi-square gate needs mvuRight, ur and the sigma table");
    for (int i = 0; i < n_q; i++) { best_idx[i] = -1; if (best_dist) best_dist[i] = 256; }
    if (n_q == 0 || S.n == 0) return IVF_OK;
    if (!q_u || !q_v || !q_radius || !q_level || !q_desc) return fail(IVF_E_INVALID, "null query array");
    if (gate)
        for (int i = 0; i < S.n; i++)
            if (S.kps[i].octave < 0 || S.kps[i].octave >= n_levels) return fail(IVF_E_INVALID, "keypoint %d: octave outside the sigma table", i);
    std::vector<int32_t> lo, hi; level_window(n_q, q_level, 1, 0, lo, hi);      // octave in [level-1, level] (:913-914)
    std::vector<int> qStart, cand, dist;
    const int rc = S.get(n_q, q_u, q_v, q_radius, lo.data(), hi.data(), q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    for (int i = 0; i < n_q; i++) {
        const float u = q_u[i], v = q_v[i], ur = q_ur ? q_ur[i] : 0.0f;
        int bestDist = 256, bestIdx = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++) {
            const int idx = cand[p];
            if (gate) {                                              // chi-square gates (:918-938), f32 products compared in double
                const ivf_keypoint& kp = S.kps[idx];
                if (S.uright[idx] >= 0) {
                    const float ex = u - kp.x, ey = v - kp.y, er = ur - S.uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kp.octave] > 7.8) continue;
                } else {
                    const float ex = u - kp.x, ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kp.octave] > 5.99) continue;
                }
            }
            if (dist[p] < bestDist) { bestDist = dist[p]; bestIdx = idx; }
        }
        if (best_dist) best_dist[i] = bestDist;
        if (bestDist <= 50) best_idx[i] = bestIdx;
    }
    return IVF_OK;
}
int ivf_fuse_candidates(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, const float* kf_uright, int n_kf,
                        const ivf_bounds* bounds, const float* inv_level_sigma2, int n_levels,
                        int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                        const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid,
                        int32_t* best_idx, int32_t* best_dist, int device_id)
{
    if (!kf_kps || !kf_desc || !bounds || !best_idx || n_kf < 0 || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    return fuse_impl(CandSource::host(kf_kps, kf_desc, kf_uright, n_kf, bounds, device_id), inv_level_sigma2, n_levels, n_q, q_u, q_v, q_ur,
                     q_radius, q_level, q_desc, q_valid, best_idx, best_dist);
}
int ivf_frame_fuse_candidates(ivf_frame* f, const float* inv_level_sigma2, int n_levels, int n_q, const float* q_u, const float* q_v,
                              const float* q_ur, const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                              const uint8_t* q_valid, int32_t* best_idx, int32_t* best_dist)
{
    if (!f || !best_idx || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    CandSource S; const int rc = CandSource::resident(f, S); if (rc) return rc;
    return fuse_impl(S, inv_level_sigma2, n_levels, n_q, q_u, q_v, q_ur, q_radius, q_level, q_desc, q_valid, best_idx, best_dist);
}

// ORBmatcher::SearchBySim3 (ORB/src/ORBmatcher.cc:1145-1254) on the two sets of projected map points
namespace {
int window_best(const CandSource& S, int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                const uint8_t* q_desc, const uint8_t* q_valid, int th, std::vector<int>& best)
{
    best.assign(n_q, -1);
    if (n_q == 0 || S.n == 0) return IVF_OK;
    std::vector<int32_t> lo, hi; level_window(n_q, q_level, 1, 0, lo, hi);      // octave in [level-1, level] (:1245-1246)
    std::vector<int> qStart, cand, dist;
    const int rc = S.get(n_q, q_u, q_v, q_radius, lo.data(), hi.data(), q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    for (int i = 0; i < n_q; i++) {
        int bestDist = INT_MAX, bestIdx = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++)
            if (dist[p] < bestDist) { bestDist = dist[p]; bestIdx = cand[p]; }
        if (bestDist <= th) best[i] = bestIdx;
    }
    return IVF_OK;
}
int sim3_impl(const CandSource& S1, const CandSource& S2,
              const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level, const uint8_t* q12_desc,
              const uint8_t* q12_valid, const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
              const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound)
{
    const int n1 = S1.n, n2 = S2.n;
    *nfound = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return IVF_OK;
    if (!q12_u || !q12_v || !q12_radius || !q12_level || !q12_desc || !q21_u || !q21_v || !q21_radius || !q21_level || !q21_desc)
        return fail(IVF_E_INVALID, "null query array");
    std::vector<int> m1, m2;                                      // vnMatch1 / vnMatch2 (:1186-1187), TH_HIGH (:1264, :1344)
    int rc = window_best(S2, n1, q12_u, q12_v, q12_radius, q12_level, q12_desc, q12_valid, 100, m1);
    if (rc) return rc;
    rc = window_best(S1, n2, q21_u, q21_v, q21_radius, q21_level, q21_desc, q21_valid, 100, m2);
    if (rc) return rc;
    int nf = 0;
    for (int i1 = 0; i1 < n1; i1++) {                             // agreement check (:1336-1349)
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { matches12[i1] = idx2; nf++; }
    }
    *nfound = nf;
    return IVF_OK;
}
}  // namespace

int ivf_search_by_sim3(const ivf_keypoint* kps1, const uint8_t* desc1, int n1, const ivf_bounds* bounds1,
                       const ivf_keypoint* kps2, const uint8_t* desc2, int n2, const ivf_bounds* bounds2,
                       const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                       const uint8_t* q12_desc, const uint8_t* q12_valid,
                       const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                       const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound, int device_id)
{
    if (!kps1 || !desc1 || !kps2 || !desc2 || !bounds1 || !bounds2 || !matches12 || !nfound || n1 < 0 || n2 < 0)
        return fail(IVF_E_INVALID, "bad argument");
    return sim3_impl(CandSource::host(kps1, desc1, nullptr, n1, bounds1, device_id), CandSource::host(kps2, desc2, nullptr, n2, bounds2, device_id),
                     q12_u, q12_v, q12_radius, q12_level, q12_desc, q12_valid, q21_u, q21_v, q21_radius, q21_level, q21_desc, q21_valid,
                     matches12, nfound);
}
int ivf_frame_search_by_sim3(ivf_frame* f1, ivf_frame* f2,
                             const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                             const uint8_t* q12_desc, const uint8_t* q12_valid,
                             const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                             const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound)
{
    if (!f1 || !f2 || !matches12 || !nfound) return fail(IVF_E_INVALID, "bad argument");
    CandSource S1, S2;
    int rc = CandSource::resident(f1, S1); if (rc) return rc;
    rc = CandSource::resident(f2, S2); if (rc) return rc;
    return sim3_impl(S1, S2, q12_u, q12_v, q12_radius, q12_level, q12_desc, q12_valid, q21_u, q21_v, q21_radius, q21_level, q21_desc,
                     q21_valid, matches12, nfound);
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORB/src/ORBmatcher.cc:165-294), feature vectors in CSR
int ivf_search_by_bow(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_map_point, int n_kf,
                      const int32_t* kf_node, const int32_t* kf_start, const int32_t* kf_idx, int kf_nodes,
                      const ivf_keypoint* f_kps, const uint8_t* f_desc, int n_f,
                      const int32_t* f_node, const int32_t* f_start, const int32_t* f_idx, int f_nodes,
                      float nn_ratio, int check_orientation, int32_t* f_match, int* nmatches, int device_id)
{
    if (!kf_kps || !kf_desc || !kf_has_map_point || !f_kps || !f_desc || !f_match || !nmatches || n_kf < 0 || n_f < 0 ||
        kf_nodes < 0 || f_nodes < 0)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    if (kf_nodes == 0 || f_nodes == 0 || n_kf == 0 || n_f == 0) return IVF_OK;
    if (!kf_node || !kf_start || !kf_idx || !f_node || !f_start || !f_idx) return fail(IVF_E_INVALID, "null feature-vector array");
    for (int a = 0; a + 1 < kf_nodes; a++) if (kf_node[a] >= kf_node[a + 1]) return fail(IVF_E_INVALID, "keyframe node ids must ascend");
    for (int b = 0; b + 1 < f_nodes; b++) if (f_node[b] >= f_node[b + 1]) return fail(IVF_E_INVALID, "frame node ids must ascend");
    for (int p = kf_start[0]; p < kf_start[kf_nodes]; p++) if (kf_idx[p] < 0 || kf_idx[p] >= n_kf) return fail(IVF_E_INVALID, "keyframe feature index out of range");
    for (int q = f_start[0]; q < f_start[f_nodes]; q++) if (f_idx[q] < 0 || f_idx[q] >= n_f) return fail(IVF_E_INVALID, "frame feature index out of range");
    // 1. node merge (:187-265): per KF feature with a map point, the run of (KF, F) pairs of its node
    struct Run { int kf, b, first; };
    std::vector<Run> runs; std::vector<int> pairs;
    {
        int a = 0, b = 0;
        while (a < kf_nodes && b < f_nodes) {
            if (kf_node[a] == f_node[b]) {
                for (int p = kf_start[a]; p < kf_start[a + 1]; p++) {
                    const int i = kf_idx[p];
                    if (!kf_has_map_point[i]) continue;
                    runs.push_back({i, b, (int)pairs.size() / 2});
                    for (int q = f_start[b]; q < f_start[b + 1]; q++) { pairs.push_back(i); pairs.push_back(f_idx[q]); }
                }
                a++; b++;
            } else if (kf_node[a] < f_node[b]) { while (a < kf_nodes && kf_node[a] < f_node[b]) a++; }
            else { while (b < f_nodes && f_node[b] < kf_node[a]) b++; }
        }
    }
    const int nPairs = (int)pairs.size() / 2;
    // 2. all in-node distances on the device
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(kf_desc, n_kf, f_desc, n_f, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    // 3. greedy replay (:200-259) and the rotation filter (:268-288)
    const int HISTO_LENGTH = 30;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0;
    for (const Run& r : runs) {
        const int len = f_start[r.b + 1] - f_start[r.b];
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int k = 0; k < len; k++) {
            const int iF = pairs[2 * (r.first + k) + 1], d = dist[r.first + k];
            if (f_match[iF] >= 0) continue;
            if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = iF; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist1 <= 50 && (float)bestDist1 < nn_ratio * (float)bestDist2) {
            f_match[bestIdxF] = r.kf;
            if (check_orientation) {
                float rot = kf_kps[r.kf].angle - f_kps[bestIdxF].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdxF);
            }
            nm++;
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { f_match[j] = -1; nm--; }
    }
    *nmatches = nm;
    return IVF_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORB/src/ORBmatcher.cc:528-661), feature vectors in CSR
int ivf_search_by_bow_keyframes(const ivf_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1, int n1,
                                const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                const ivf_keypoint* kps2, const uint8_t* desc2, const uint8_t* has_map_point2, int n2,
                                const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches, int device_id)
{
    if (!kps1 || !desc1 || !has_map_point1 || !kps2 || !desc2 || !has_map_point2 || !matches12 || !nmatches || n1 < 0 || n2 < 0 ||
        nodes1 < 0 || nodes2 < 0)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (nodes1 == 0 || nodes2 == 0 || n1 == 0 || n2 == 0) return IVF_OK;
    if (!node1 || !start1 || !idx1 || !node2 || !start2 || !idx2) return fail(IVF_E_INVALID, "null feature-vector array");
    for (int a = 0; a + 1 < nodes1; a++) if (node1[a] >= node1[a + 1]) return fail(IVF_E_INVALID, "node ids of keyframe 1 must ascend");
    for (int b = 0; b + 1 < nodes2; b++) if (node2[b] >= node2[b + 1]) return fail(IVF_E_INVALID, "node ids of keyframe 2 must ascend");
    for (int p = start1[0]; p < start1[nodes1]; p++) if (idx1[p] < 0 || idx1[p] >= n1) return fail(IVF_E_INVALID, "feature index of keyframe 1 out of range");
    for (int q = start2[0]; q < start2[nodes2]; q++) if (idx2[q] < 0 || idx2[q] >= n2) return fail(IVF_E_INVALID, "feature index of keyframe 2 out of range");
    struct Run { int i1, b, first; };
    std::vector<Run> runs; std::vector<int> pairs;
    {
        int a = 0, b = 0;
        while (a < nodes1 && b < nodes2) {
            if (node1[a] == node2[b]) {
                for (int p = start1[a]; p < start1[a + 1]; p++) {
                    const int i = idx1[p];
                    if (!has_map_point1[i]) continue;
                    runs.push_back({i, b, (int)pairs.size() / 2});
                    for (int q = start2[b]; q < start2[b + 1]; q++) { pairs.push_back(i); pairs.push_back(idx2[q]); }
                }
                a++; b++;
            } else if (node1[a] < node2[b]) { while (a < nodes1 && node1[a] < node2[b]) a++; }
            else { while (b < nodes2 && node2[b] < node1[a]) b++; }
        }
    }
    const int nPairs = (int)pairs.size() / 2;
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(desc1, n1, desc2, n2, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    const int HISTO_LENGTH = 30;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> matched2(n2, 0);
    int nm = 0;
    for (const Run& r : runs) {
        const int len = start2[r.b + 1] - start2[r.b];
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int k = 0; k < len; k++) {
            const int i2 = pairs[2 * (r.first + k) + 1], d = dist[r.first + k];
            if (matched2[i2] || !has_map_point2[i2]) continue;
            if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist1 < 50 && (float)bestDist1 < nn_ratio * (float)bestDist2) {          // strict '<' here (:598)
            matches12[r.i1] = bestIdx2; matched2[bestIdx2] = 1;
            if (check_orientation) {
                float rot = kps1[r.i1].angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(r.i1);
            }
            nm++;
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { matches12[j] = -1; nm--; }
    }
    *nmatches = nm;
    return IVF_OK;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) (ORB/src/ORBmatcher.cc:1520-1652)
static int reloc_impl(const CandSource& S, int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                      const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid, int orb_dist, int check_orientation,
                      int32_t* cur_assign, int* nmatches)
{
    const ivf_keypoint* cur_kps = S.kps;
    *nmatches = 0;
    if (n_q == 0 || S.n == 0) return IVF_OK;
    if (!q_u || !q_v || !q_radius || !q_level || !q_angle || !q_desc) return fail(IVF_E_INVALID, "null query array");
    std::vector<int32_t> lo, hi; level_window(n_q, q_level, 1, 1, lo, hi);      // GetFeaturesInArea(u, v, radius, level-1, level+1) (:1574)
    std::vector<int> qStart, cand, dist;
    int rc = S.get(n_q, q_u, q_v, q_radius, lo.data(), hi.data(), q_desc, q_valid, qStart, cand, dist);
    if (rc) return rc;
    const int HISTO_LENGTH = 30;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < n_q; i++) {
        int bestDist = 256, bestIdx2 = -1;
        for (int p = qStart[i]; p < qStart[i + 1]; p++) {
            const int i2 = cand[p];
            if (cur_assign[i2] != -1) continue;
            if (dist[p] < bestDist) { bestDist = dist[p]; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist && bestIdx2 >= 0) {
            cur_assign[bestIdx2] = i; nm++;
            if (check_orientation) {
                float rot = q_angle[i] - cur_kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { cur_assign[j] = -1; nm--; }
    }
    *nmatches = nm;
    return IVF_OK;
}

int ivf_search_by_projection_reloc(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, int n_cur, const ivf_bounds* bounds,
                                   int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                   const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                   int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches, int device_id)
{
    if (!cur_kps || !cur_desc || !bounds || !cur_assign || !nmatches || n_cur < 0 || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    return reloc_impl(CandSource::host(cur_kps, cur_desc, nullptr, n_cur, bounds, device_id), n_q, q_u, q_v, q_radius, q_level, q_angle,
                      q_desc, q_valid, orb_dist, check_orientation, cur_assign, nmatches);
}
int ivf_frame_search_by_projection_reloc(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                                         const int32_t* q_level, const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                         int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches)
{
    if (!f || !cur_assign || !nmatches || n_q < 0) return fail(IVF_E_INVALID, "bad argument");
    CandSource S; const int rc = CandSource::resident(f, S); if (rc) return rc;
    return reloc_impl(S, n_q, q_u, q_v, q_radius, q_level, q_angle, q_desc, q_valid, orb_dist, check_orientation, cur_assign, nmatches);
}

// ORBmatcher::SearchForTriangulation (ORB/src/ORBmatcher.cc:663-829) + CheckDistEpipolarLine (:146-163)
int ivf_search_for_triangulation(const ivf_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1, const uint8_t* stereo1, int n1,
                                 const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                 const ivf_keypoint* kps2, const uint8_t* desc2, const uint8_t* has_map_point2, const uint8_t* stereo2, int n2,
                                 const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int n_levels,
                                 int only_stereo, int check_orientation, int32_t* matches12, int* nmatches, int device_id)
{
    if (!kps1 || !desc1 || !has_map_point1 || !stereo1 || !kps2 || !desc2 || !has_map_point2 || !stereo2 || !F12 || !scale_factors2 ||
        !level_sigma2_2 || !matches12 || !nmatches || n1 < 0 || n2 < 0 || nodes1 < 0 || nodes2 < 0 || n_levels < 1)
        return fail(IVF_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (nodes1 == 0 || nodes2 == 0 || n1 == 0 || n2 == 0) return IVF_OK;
    if (!node1 || !start1 || !idx1 || !node2 || !start2 || !idx2) return fail(IVF_E_INVALID, "null feature-vector array");
    for (int a = 0; a + 1 < nodes1; a++) if (node1[a] >= node1[a + 1]) return fail(IVF_E_INVALID, "node ids of keyframe 1 must ascend");
    for (int b = 0; b + 1 < nodes2; b++) if (node2[b] >= node2[b + 1]) return fail(IVF_E_INVALID, "node ids of keyframe 2 must ascend");
    for (int p = start1[0]; p < start1[nodes1]; p++) if (idx1[p] < 0 || idx1[p] >= n1) return fail(IVF_E_INVALID, "feature index of keyframe 1 out of range");
    for (int q = start2[0]; q < start2[nodes2]; q++) if (idx2[q] < 0 || idx2[q] >= n2) return fail(IVF_E_INVALID, "feature index of keyframe 2 out of range");
    for (int i = 0; i < n2; i++) if (kps2[i].octave < 0 || kps2[i].octave >= n_levels) return fail(IVF_E_INVALID, "keypoint %d of keyframe 2: octave outside the tables", i);
    // 1. node merge: per eligible KF1 feature the run of eligible (i1, i2) pairs of its node (:697-731)
    struct Run { int i1, first, len; };
    std::vector<Run> runs; std::vector<int> pairs;
    {
        int a = 0, b = 0;
        while (a < nodes1 && b < nodes2) {
            if (node1[a] == node2[b]) {
                for (int p = start1[a]; p < start1[a + 1]; p++) {
                    const int i1 = idx1[p];
                    if (has_map_point1[i1] || (only_stereo && !stereo1[i1])) continue;
                    Run r{i1, (int)pairs.size() / 2, 0};
                    for (int q = start2[b]; q < start2[b + 1]; q++) {
                        const int i2 = idx2[q];
                        if (has_map_point2[i2] || (only_stereo && !stereo2[i2])) continue;
                        pairs.push_back(i1); pairs.push_back(i2); r.len++;
                    }
                    runs.push_back(r);
                }
                a++; b++;
            } else if (node1[a] < node2[b]) { while (a < nodes1 && node1[a] < node2[b]) a++; }
            else { while (b < nodes2 && node2[b] < node1[a]) b++; }
        }
    }
    const int nPairs = (int)pairs.size() / 2;
    std::vector<int> dist(std::max(nPairs, 1));
    int rc = ivf_hamming_pairs(desc1, n1, desc2, n2, pairs.data(), nPairs, dist.data(), device_id);
    if (rc) return rc;
    // 2. replay with the epipole and epipolar-line gates (f32 arithmetic as written in :149-162, compared in double)
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0;
    for (const Run& r : runs) {
        const ivf_keypoint& kp1 = kps1[r.i1];
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int k = 0; k < r.len; k++) {
            const int i2 = pairs[2 * (r.first + k) + 1], d = dist[r.first + k];
            if (d > TH_LOW || d > bestDist) continue;
            const ivf_keypoint& kp2 = kps2[i2];
            if (!stereo1[r.i1] && !stereo2[i2]) {
                const float distex = ex - kp2.x, distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * scale_factors2[kp2.octave]) continue;
            }
            const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
            const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
            const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
            const float num = a * kp2.x + b * kp2.y + c;
            const float den = a * a + b * b;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (dsqr < 3.84 * level_sigma2_2[kp2.octave]) { bestIdx2 = i2; bestDist = d; }
        }
        if (bestIdx2 >= 0) {
            matches12[r.i1] = bestIdx2; nm++;
            if (check_orientation) {
                float rot = kp1.angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(r.i1);
            }
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { matches12[j] = -1; nm--; }
    }
    *nmatches = nm;
    return IVF_OK;
}

// DBoW2 vocabulary (device-resident tree) and TemplatedVocabulary::transform per descriptor
struct ivf_vocabulary {
    int device = 0, nNodes = 0, depth = 0, maxChildren = 0;
    int *dChildStart = nullptr, *dChild = nullptr; uint8_t* dDesc = nullptr;
    std::vector<int> word; std::vector<double> weight; std::vector<int> childStart;
};

int ivf_vocabulary_create(int n_nodes, const int32_t* child_start, const int32_t* child, const uint8_t* node_desc,
                          const int32_t* node_word, const double* node_weight, int depth_L, int device_id, ivf_vocabulary** out)
{
    if (!out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (n_nodes < 2 || !child_start || !child || !node_desc || !node_word || !node_weight || depth_L < 1)
        return fail(IVF_E_INVALID, "bad argument");
    if (child_start[0] != 0) return fail(IVF_E_INVALID, "child_start[0] must be 0");
    const int nChild = child_start[n_nodes];
    if (child_start[1] == child_start[0]) return fail(IVF_E_INVALID, "the root (node 0) has no children");
    int maxC = 0;
    for (int i = 0; i < n_nodes; i++) {
        if (child_start[i + 1] < child_start[i]) return fail(IVF_E_INVALID, "child_start must not decrease (node %d)", i);
        maxC = std::max(maxC, child_start[i + 1] - child_start[i]);
    }
    if (maxC > 65535) return fail(IVF_E_INVALID, "more than 65535 children under one node");
    for (int c = 0; c < nChild; c++) if (child[c] <= 0 || child[c] >= n_nodes) return fail(IVF_E_INVALID, "child %d: node id out of range", c);
    {   // the descent kernel loops until it meets a leaf: refuse anything that is not a tree rooted at node 0
        std::vector<char> seen(n_nodes, 0); std::vector<int> stack{0}; seen[0] = 1;
        while (!stack.empty()) {
            const int i = stack.back(); stack.pop_back();
            for (int c = child_start[i]; c < child_start[i + 1]; c++) {
                if (seen[child[c]]) return fail(IVF_E_INVALID, "node %d is reachable twice: not a tree", child[c]);
                seen[child[c]] = 1; stack.push_back(child[c]);
            }
        }
    }
    int rc = have_device(device_id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_id));
    ivf_vocabulary* v = new ivf_vocabulary();
    v->device = device_id; v->nNodes = n_nodes; v->depth = depth_L; v->maxChildren = maxC;
    v->word.assign(node_word, node_word + n_nodes); v->weight.assign(node_weight, node_weight + n_nodes);
    v->childStart.assign(child_start, child_start + n_nodes + 1);
    if (hipMalloc(&v->dChildStart, (size_t)(n_nodes + 1) * sizeof(int)) != hipSuccess || hipMalloc(&v->dChild, (size_t)std::max(nChild, 1) * sizeof(int)) != hipSuccess ||
        hipMalloc(&v->dDesc, (size_t)n_nodes * 32) != hipSuccess) { ivf_vocabulary_destroy(v); return fail(IVF_E_NO_DEVICE, "hipMalloc failed for the vocabulary"); }
    if (hipMemcpy(v->dChildStart, child_start, (size_t)(n_nodes + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->dChild, child, (size_t)nChild * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->dDesc, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice) != hipSuccess) {
        ivf_vocabulary_destroy(v);
        return fail(IVF_E_NO_DEVICE, "vocabulary upload failed");
    }
    *out = v;
    return IVF_OK;
}

void ivf_vocabulary_destroy(ivf_vocabulary* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->dChildStart) (void)hipFree(v->dChildStart);
    if (v->dChild) (void)hipFree(v->dChild);
    if (v->dDesc) (void)hipFree(v->dDesc);
    delete v;
}

int ivf_bow_transform(const ivf_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight)
{
    if (!v || n < 0 || (n > 0 && (!desc || !word_id || !node_id || !weight))) return fail(IVF_E_INVALID, "bad argument");
    if (n == 0) return IVF_OK;
    HIPCHK(hipSetDevice(v->device));
    uint8_t* scb = nullptr;
    const int src = thread_scratch(v->device, (size_t)n * 32 + 256 + (size_t)n * 2 * sizeof(int), &scb);
    if (src) return src;
    uint8_t* dD = scb; int* dOut = (int*)(scb + (((size_t)n * 32 + 255) & ~(size_t)255));
    HIPCHK(hipMemcpyAsync(dD, desc, (size_t)n * 32, hipMemcpyHostToDevice, nullptr));
    launch_bow_transform(v->dChildStart, v->dChild, v->dDesc, dD, n, v->depth - levelsup, dOut, dOut + n, nullptr);
    HIPCHK(hipGetLastError());
    std::vector<int> res((size_t)n * 2);
    HIPCHK(hipMemcpy(res.data(), dOut, (size_t)n * 2 * sizeof(int), hipMemcpyDeviceToHost));
    for (int f = 0; f < n; f++) {
        const int leaf = res[f];
        word_id[f] = v->word[leaf]; weight[f] = v->weight[leaf]; node_id[f] = res[(size_t)n + f];
    }
    return IVF_OK;
}

// BowVector / FeatureVector of one frame from the per-descriptor results (TemplatedVocabulary.h:1126-1204 with TF_IDF weights
// and L1 normalisation, the ORB vocabulary's settings; BowVector.cpp:34-46, 62-84; FeatureVector.cpp:31-45)
int ivf_bow_vectors(const int32_t* word_id, const int32_t* node_id, const double* weight, int n,
                    int32_t* bow_word, double* bow_value, int bow_cap, int* bow_n,
                    int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int fv_cap, int* fv_n)
{
    if (n < 0 || !bow_n || !fv_n || (n > 0 && (!word_id || !node_id || !weight))) return fail(IVF_E_INVALID, "bad argument");
    std::map<int, double> bow; std::map<int, std::vector<int>> fv;
    for (int f = 0; f < n; f++) {
        if (!(weight[f] > 0)) continue;                                  // stopped word (:1157)
        bow[word_id[f]] += weight[f];                                    // addWeight
        fv[node_id[f]].push_back(f);                                     // addFeature
    }
    double norm = 0.0;
    for (auto& kv : bow) norm += fabs(kv.second);                        // L1 (BowVector.cpp:67-71)
    if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
    *bow_n = (int)bow.size(); *fv_n = (int)fv.size();
    if ((int)bow.size() > bow_cap || (int)fv.size() > fv_cap) return fail(IVF_E_CAPACITY, "%zu words / %zu nodes exceed the capacities", bow.size(), fv.size());
    int k = 0;
    for (auto& kv : bow) { if (bow_word) bow_word[k] = kv.first; if (bow_value) bow_value[k] = kv.second; k++; }
    k = 0; int pos = 0;
    if (fv_start) fv_start[0] = 0;
    for (auto& kv : fv) {
        if (fv_node) fv_node[k] = kv.first;
        for (int i : kv.second) { if (fv_idx) fv_idx[pos] = i; pos++; }
        if (fv_start) fv_start[k + 1] = pos;
        k++;
    }
    return IVF_OK;
}

// MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312): all-pairs Hamming + row medians on the device,
// first minimum on the host
int ivf_distinctive_descriptor(const uint8_t* desc, int n, int* best_index, int* best_median, int device_id)
{
    if (!desc || !best_index || n < 1) return fail(IVF_E_INVALID, "bad argument");
    int rc = have_device(device_id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_id));
    uint8_t* scb = nullptr;
    rc = thread_scratch(device_id, (size_t)n * 32 + 256 + (size_t)n * sizeof(int), &scb);
    if (rc) return rc;
    uint8_t* dD = scb; int* dM = (int*)(scb + (((size_t)n * 32 + 255) & ~(size_t)255));
    HIPCHK(hipMemcpyAsync(dD, desc, (size_t)n * 32, hipMemcpyHostToDevice, nullptr));
    launch_distinct_median(dD, n, dM, nullptr);
    HIPCHK(hipGetLastError());
    std::vector<int> med(n);
    HIPCHK(hipMemcpy(med.data(), dM, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    int bm = INT_MAX, bi = 0;
    for (int i = 0; i < n; i++) if (med[i] < bm) { bm = med[i]; bi = i; }
    *best_index = bi;
    if (best_median) *best_median = bm;
    return IVF_OK;
}

// ORBmatcher::UpdateQualityScores(Frame &F) (ORB/src/ORBmatcher.cc:1108-1121): host bookkeeping, sequential by definition
int ivf_update_quality_scores(const int32_t* assign, int n, float* kp_quality, float* mp_quality, int n_map_points)
{
    if (!assign || !kp_quality || !mp_quality || n < 0 || n_map_points < 0) return fail(IVF_E_INVALID, "bad argument");
    const float kDeltaThresh = 0.01f;
    for (int i = 0; i < n; i++) {
        const int m = assign[i];
        if (m < 0) continue;
        if (m >= n_map_points) return fail(IVF_E_INVALID, "assign[%d] = %d outside the %d map points", i, m, n_map_points);
        const float mpt = mp_quality[m];
        const float upd = std::min(mpt, kp_quality[i]);
        if (fabsf(upd - mpt) > kDeltaThresh) mp_quality[m] = upd;
        kp_quality[i] = upd;
    }
    return IVF_OK;
}

// testing hook (see include/ivfront.h)
int ivf_test_retain_best(const float* responses, int n, int n_points, int32_t* order_out, int device_id)
{
    if (!responses || !order_out || n < 1 || n > 4096 || n_points < 0) return fail(IVF_E_INVALID, "bad argument (1 <= n <= 4096)");
    int rc = have_device(device_id);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device_id));
    uint8_t* scb = nullptr;
    rc = thread_scratch(device_id, (size_t)n * 8 + 256, &scb);
    if (rc) return rc;
    float* dR = (float*)scb; int* dO = (int*)(scb + (((size_t)n * 4 + 255) & ~(size_t)255));
    HIPCHK(hipMemcpyAsync(dR, responses, (size_t)n * sizeof(float), hipMemcpyHostToDevice, nullptr));
    launch_test_retain_best(dR, n, n_points, dO, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(order_out, dO, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return IVF_OK;
}

// ---- batched stereo front end ----
// The kPipe internal streams of a device: created on first use, shared by every ivf_frontend on that device, never destroyed (no HIP
// call during static destruction).  The tail kernels of a batch are small and latency-bound: on high-priority streams they slot in
// beside whatever large kernels the caller's stream is running (the FCN of the next batch) instead of queueing behind them.
static int internal_streams(int device, hipStream_t (&out)[kPipe])
{
    struct Set { hipStream_t s[kPipe]; };
    static std::mutex* m = new std::mutex();
    static std::map<int, Set>* sets = new std::map<int, Set>();
    std::lock_guard<std::mutex> g(*m);
    auto it = sets->find(device);
    if (it == sets->end()) {
        HIPCHK(hipSetDevice(device));
        int prLo = 0, prHi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prLo, &prHi);
        static const bool noPrio = IVF_EXP_ENV("IVF_NO_STREAM_PRIORITY") != nullptr;
        Set st{};
        for (int k = 0; k < kPipe; k++)
            if ((noPrio ? hipStreamCreateWithFlags(&st.s[k], hipStreamNonBlocking)
                        : hipStreamCreateWithPriority(&st.s[k], hipStreamNonBlocking, prHi)) != hipSuccess) {
                for (int j = 0; j < k; j++) (void)hipStreamDestroy(st.s[j]);
                return fail(IVF_E_NO_DEVICE, "stream creation failed");
            }
        it = sets->emplace(device, st).first;
    }
    for (int k = 0; k < kPipe; k++) out[k] = it->second.s[k];
    return IVF_OK;
}

int ivf_frontend_create(const ivf_frontend_config* cfg, ivf_frontend** out)
{
    if (!cfg || !out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    const ivf_extractor_params &L = cfg->left, &R = cfg->right;
    if (L.nfeatures != R.nfeatures || L.scale_factor != R.scale_factor || L.nlevels != R.nlevels ||
        L.ini_th_fast != R.ini_th_fast || L.min_th_fast != R.min_th_fast)
        return fail(IVF_E_INVALID, "left/right extractor parameters may differ only in enable_introspection");
    if (cfg->max_pairs < 1) return fail(IVF_E_INVALID, "max_pairs must be >= 1");
    if (!(cfg->b > 0.f) || !(cfg->bf > 0.f)) return fail(IVF_E_INVALID, "bf and b must be positive");
    Tables t;
    ivf_extractor_params p = L;
    p.enable_introspection = (L.enable_introspection || R.enable_introspection) ? 1 : 0;
    int rc = make_tables(p, t);
    if (rc) return rc;
    rc = have_device(cfg->device_id);
    if (rc) return rc;
    ivf_frontend* fe = new ivf_frontend();
    fe->cfg = *cfg; fe->tl = t;
    auto cleanup = [&](int code) { ivf_frontend_destroy(fe); return code; };
    for (int k = 0; k < kPipe; k++) {
        rc = fe->ctx[k].build(t, cfg->width, cfg->height, 2 * cfg->max_pairs, 2, cfg->device_id, true);
        if (rc) return cleanup(rc);
        fe->ctx[k].markOwn = k; fe->ctx[k].markSide = 3 + (k + 1) % kPipe;
        // r05: the kPipe internal streams come from a process-wide pool per device (internal_streams above) and are SHARED by every
        // front end on that device.  Streams are a scarce resource on this runtime -- it multiplexes them onto a few hardware queues
        // (three more streams cost configs[2] 15 % just by existing) -- so a second front end in the process
        // (bench.py's configs[1] leg beside the configs[2] one; Tracking's left / right / initialisation extractors) must not add any,
        // and a create-run-destroy loop no longer creates and destroys streams at all.  Two front ends that share streams serialise
        // against each other on them, which is what they would do on the GPU anyway.
        hipStream_t pool[kPipe];
        rc = internal_streams(cfg->device_id, pool);
        if (rc) return cleanup(rc);
        fe->stream[k] = pool[k];
        const unsigned evFlags = hipEventDisableTiming;
        if (hipEventCreateWithFlags(&fe->evIn[k], evFlags) != hipSuccess ||
            hipEventCreateWithFlags(&fe->evConsumed[k], evFlags) != hipSuccess ||
            hipEventCreateWithFlags(&fe->evDone[k], evFlags) != hipSuccess)
            return cleanup(fail(IVF_E_NO_DEVICE, "event creation failed"));
    }
    std::vector<uint8_t> flags(2 * (size_t)cfg->max_pairs);
    // bit 0: the cost pyramid gates this image's extraction; bit 1: mvKeyQualScore of the LEFT keypoints reads the cost image
    // whenever one is passed, whatever the extractor flags say (Frame.cc:130-143 tests only !imDepth.empty())
    for (int i = 0; i < cfg->max_pairs; i++) { flags[2 * i] = (L.enable_introspection ? 1 : 0) | 2; flags[2 * i + 1] = R.enable_introspection ? 1 : 0; }
    if (hipMalloc(&fe->dFlags, flags.size()) != hipSuccess ||
        hipMemcpy(fe->dFlags, flags.data(), flags.size(), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(IVF_E_NO_DEVICE, "flag upload failed"));
    *out = fe;
    return IVF_OK;
}

void ivf_frontend_destroy(ivf_frontend* fe)
{
    if (!fe) return;
    (void)hipSetDevice(fe->cfg.device_id);
    // r06: wait for THIS handle's batches (their completion events), not for the device: the internal streams are shared by every front end of
    // the process on this device (internal_streams), so a device-wide wait made one camera's tear-down wait for the other's work.  Work the caller
    // put behind a batch on a lent stream (ivf_frontend_batch_stream) is the caller's to finish first, as ivfront.h says; the buffer releases below
    // go through hipFree, which itself does not return while the device still uses the allocation.
    for (int k = 0; k < kPipe; k++)
        if (fe->evDone[k]) (void)hipEventSynchronize(fe->evDone[k]);      // never recorded = complete
    if (fe->dFlags) (void)hipFree(fe->dFlags);
    for (int k = 0; k < kPipe; k++) {
        fe->ctx[k].release();
        // fe->stream[k] belongs to the process-wide pool: never destroyed
        if (fe->evIn[k]) (void)hipEventDestroy(fe->evIn[k]);
        if (fe->evConsumed[k]) (void)hipEventDestroy(fe->evConsumed[k]);
        if (fe->evDone[k]) (void)hipEventDestroy(fe->evDone[k]);
    }
    delete fe;
}

static int frontend_run_common(ivf_frontend* fe, const uint8_t* d_left, const uint8_t* d_right, const uint8_t* d_cost,
                               size_t image_stride, int row_stride, int n_pairs, void* hip_stream, const Context::SideSrc* sides);

int ivf_frontend_run(ivf_frontend* fe, const uint8_t* d_left, const uint8_t* d_right, const uint8_t* d_cost,
                     size_t image_stride, int row_stride, int n_pairs, void* hip_stream)
{
    if (!fe || !d_left || !d_right) return fail(IVF_E_INVALID, "null argument");
    if (n_pairs < 1 || n_pairs > fe->cfg.max_pairs) return fail(IVF_E_INVALID, "n_pairs %d outside [1,%d]", n_pairs, fe->cfg.max_pairs);
    if (row_stride < fe->cfg.width || image_stride < (size_t)row_stride * (fe->cfg.height - 1) + fe->cfg.width)
        return fail(IVF_E_INVALID, "strides too small for %dx%d", fe->cfg.width, fe->cfg.height);
    return frontend_run_common(fe, d_left, d_right, d_cost, image_stride, row_stride, n_pairs, hip_stream, nullptr);
}

int ivf_frontend_cost_plane(ivf_frontend* fe, uint8_t** d_plane, size_t* image_stride, int* row_stride, void* hip_stream)
{
    if (!fe || !d_plane || !image_stride || !row_stride) return fail(IVF_E_INVALID, "null argument");
    const int k = (int)(fe->runs % kPipe);
    Context& c = fe->ctx[k];
    HIPCHK(hipSetDevice(fe->cfg.device_id));
    if (!c.b.qpyr) {            // extractors that ignore the map (enableIntrospection = 0) allocate the plane on first use
        const size_t blob = (size_t)c.hc.pyrBytes * c.maxImg;
        HIPCHK(hipMalloc(&c.b.qpyr, blob));
        HIPCHK(hipMemset(c.b.qpyr, 0, blob));
    }
    // the context's previous batch (three runs ago) read this plane: the caller's stream may write it once that batch is done
    if (fe->runs >= kPipe) HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, fe->evDone[k], 0));
    *d_plane = c.b.qpyr + c.hc.lv[0].off;
    *image_stride = (size_t)2 * c.hc.pyrBytes;
    *row_stride = c.hc.lv[0].pitch;
    return IVF_OK;
}

int ivf_frontend_run_color(ivf_frontend* fe, const uint8_t* d_left, int left_code, size_t left_image_stride, int left_row_stride,
                           const uint8_t* d_right, int right_code, size_t right_image_stride, int right_row_stride,
                           const uint8_t* d_cost, size_t cost_image_stride, int cost_row_stride, int n_pairs, void* hip_stream)
{
    if (!fe || !d_left || !d_right) return fail(IVF_E_INVALID, "null argument");
    if (n_pairs < 1 || n_pairs > fe->cfg.max_pairs) return fail(IVF_E_INVALID, "n_pairs %d outside [1,%d]", n_pairs, fe->cfg.max_pairs);
    Context::SideSrc sides[2];
    const uint8_t* ptr[2] = {d_left, d_right}; const int code[2] = {left_code, right_code};
    const size_t ist[2] = {left_image_stride, right_image_stride}; const int rst[2] = {left_row_stride, right_row_stride};
    for (int sd = 0; sd < 2; sd++) {
        if (code[sd] < 0 || (code[sd] & 3) == 3 || code[sd] > 7 || (!(code[sd] & 3) && code[sd]))
            return fail(IVF_E_INVALID, "side %d: colour code %d is not 0 (grey), 1 (bytes B,G,R) or 2 (bytes R,G,B), the last two optionally + 4 (OpenCV <= 3 coefficients)", sd, code[sd]);
        const int ch = (code[sd] & 3) ? 3 : 1;
        if (rst[sd] < ch * fe->cfg.width || ist[sd] < (size_t)rst[sd] * (fe->cfg.height - 1) + (size_t)ch * fe->cfg.width)
            return fail(IVF_E_INVALID, "side %d: strides too small for %dx%d x %d channel(s)", sd, fe->cfg.width, fe->cfg.height, ch);
        sides[sd].src = ptr[sd]; sides[sd].imageStride = ist[sd]; sides[sd].rowStride = rst[sd]; sides[sd].code = code[sd];
    }
    if (d_cost && (cost_row_stride < fe->cfg.width || cost_image_stride < (size_t)cost_row_stride * (fe->cfg.height - 1) + fe->cfg.width))
        return fail(IVF_E_INVALID, "cost strides too small for %dx%d", fe->cfg.width, fe->cfg.height);
    return frontend_run_common(fe, d_left, d_right, d_cost, cost_image_stride, cost_row_stride, n_pairs, hip_stream, sides);
}

// sides != nullptr: the images come from sides[0 / 1] (own strides, grey or colour); image_stride / row_stride then describe the cost maps only
static int frontend_run_common(ivf_frontend* fe, const uint8_t* d_left, const uint8_t* d_right, const uint8_t* d_cost,
                               size_t image_stride, int row_stride, int n_pairs, void* hip_stream, const Context::SideSrc* sides)
{
    hipStream_t caller = (hipStream_t)hip_stream;
    const int k = (int)(fe->runs % kPipe);
    Context& c = fe->ctx[k];
    if (sides) { c.sideSrc[0] = sides[0]; c.sideSrc[1] = sides[1]; }
    hipStream_t st = fe->stream[k];
    HIPCHK(hipSetDevice(fe->cfg.device_id));
    // order: everything the caller enqueued so far (it produced the inputs) -> this batch
    IVF_MARK_HOST(6, (fe->runs << 8) | 1);
    HIPCHK(hipEventRecord(fe->evIn[k], caller));
    HIPCHK(hipStreamWaitEvent(st, fe->evIn[k], 0));
    // the blur of this batch is lent the internal stream of the NEXT context: whatever older batch that stream still holds does not
    // depend on this one, and the next batch queues behind the blur
    int rc = c.run(d_left, d_right, d_cost, image_stride, row_stride, image_stride, row_stride, 2 * n_pairs,
                   fe->dFlags, st, fe->evConsumed[k], fe->stream[(k + 1) % kPipe]);
    if (rc) return rc;
    launch_stereo(c.hc, c.dc, c.b, n_pairs, fe->cfg.bf, fe->cfg.b, st);
    IVF_MARK(st, c.markOwn, 7, c.nRuns - 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(fe->evDone[k], st));
    // the caller's stream may overwrite its input buffers once they have been ingested
    HIPCHK(hipStreamWaitEvent(caller, fe->evConsumed[k], 0));
    fe->lastPairs = n_pairs;
    fe->pairsOf[k] = n_pairs;
    IVF_MARK_HOST(6, (fe->runs << 8) | 2);
    fe->runs++;
    return IVF_OK;
}

int ivf_frontend_sync(ivf_frontend* fe)
{
    if (!fe) return fail(IVF_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(fe->cfg.device_id));
    for (int k = 0; k < kPipe; k++) {
        IVF_MARK_HOST(7, 0x10 + k);
        HIPCHK(hipStreamSynchronize(fe->stream[k]));
        const int rc = fe->ctx[k].check_status(k);
        if (rc) return rc;
    }
    IVF_MARK_HOST(7, 0x20);
    return IVF_OK;
}

#ifdef IVF_EXPERIMENT
// soak aid, experiment builds only (not declared in include/ivfront.h): `words` = at least 8 ints of host memory the CALLER keeps
// alive (e.g. a MAP_SHARED file mapping its parent process also maps); registered with the runtime here, written by the marker
// launches from then on.  words == NULL switches the markers off again.
extern "C" __attribute__((visibility("default"))) int ivf_debug_progress_words(int* words, int n_words, int device_id)
{
    if (!words) { g_markDev = nullptr; g_markHost = nullptr; return IVF_OK; }
    if (n_words < 8) return fail(IVF_E_INVALID, "need at least 8 progress words");
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipHostRegister(words, (size_t)n_words * sizeof(int), hipHostRegisterMapped));
    void* d = nullptr;
    HIPCHK(hipHostGetDevicePointer(&d, words, 0));
    g_markDev = (int*)d; g_markHost = words;
    return IVF_OK;
}
#endif

int ivf_frontend_device_results(const ivf_frontend* fe, int side, const ivf_keypoint** d_kps, const uint8_t** d_desc,
                                const int32_t** d_count, const float** d_uright, const float** d_depth,
                                const float** d_quality, int* cap)
{
    if (!fe || side < 0 || side > 1) return fail(IVF_E_INVALID, "bad argument");
    const Buffers& b = fe->ctx[fe->last()].b;
    const size_t nf = fe->ctx[0].hc.nfeatures;
    // images are interleaved [L0,R0,L1,R1,...]: element stride between pairs is 2*cap
    if (d_kps) *d_kps = b.kps + side * nf;
    if (d_desc) *d_desc = b.desc + side * nf * 32;
    if (d_count) *d_count = b.count + side;
    if (d_uright) *d_uright = side == 0 ? b.uright : nullptr;
    if (d_depth) *d_depth = side == 0 ? b.depth : nullptr;
    if (d_quality) *d_quality = b.quality + side * nf;
    if (cap) *cap = (int)nf;
    return IVF_OK;
}

int ivf_frontend_fetch_of(ivf_frontend* fe, int age, int pair, int side, ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                          float* uright, float* depth, float* quality)
{
    if (!fe || !n_out || side < 0 || side > 1) return fail(IVF_E_INVALID, "bad argument");
    if (age < 0 || age >= kPipe || fe->runs <= age) return fail(IVF_E_STATE, "no batch of age %d is held", age);
    const int k = (int)((fe->runs - 1 - age) % kPipe);
    if (pair < 0 || pair >= fe->pairsOf[k]) return fail(IVF_E_INVALID, "pair %d outside the batch of %d", pair, fe->pairsOf[k]);
    int rc = ivf_frontend_sync(fe);
    if (rc) return rc;
    const Buffers& b = fe->ctx[k].b;
    const size_t nf = fe->ctx[0].hc.nfeatures, img = (size_t)pair * 2 + side;
    int n = 0;
    HIPCHK(hipMemcpy(&n, b.count + img, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) return fail(IVF_E_CAPACITY, "%d keypoints exceed caller capacity %d", n, cap);
    if (n == 0) return IVF_OK;
    if (kps) HIPCHK(hipMemcpy(kps, b.kps + img * nf, (size_t)n * sizeof(ivf_keypoint), hipMemcpyDeviceToHost));
    if (desc) HIPCHK(hipMemcpy(desc, b.desc + img * nf * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (quality) HIPCHK(hipMemcpy(quality, b.quality + img * nf, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (side == 0) {
        if (uright) HIPCHK(hipMemcpy(uright, b.uright + (size_t)pair * nf, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (depth) HIPCHK(hipMemcpy(depth, b.depth + (size_t)pair * nf, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return IVF_OK;
}

int ivf_frontend_fetch(ivf_frontend* fe, int pair, int side, ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                       float* uright, float* depth, float* quality)
{
    if (fe && fe->runs == 0) return fail(IVF_E_INVALID, "pair %d outside the last batch of 0", pair);
    return ivf_frontend_fetch_of(fe, 0, pair, side, kps, desc, cap, n_out, uright, depth, quality);
}

float ivf_frontend_last_fast_ms(ivf_frontend* fe)
{
    double sum = 0; int n = 0;
    if (ivf_frontend_fast_ms_stats(fe, 1, &sum, &n) != IVF_OK || n < 1) return -1.f;
    return (float)sum;
}

int ivf_frontend_fast_ms_stats(ivf_frontend* fe, int last_n, double* sum_ms, int* n_out)
{
    if (!fe || !sum_ms || !n_out) return fail(IVF_E_INVALID, "null argument");
    *sum_ms = 0; *n_out = 0;
    HIPCHK(hipSetDevice(fe->cfg.device_id));
    // run r used context r % kPipe and that context's ring slot (r / kPipe) % kEvRing
    const long long keep = kPipe * (long long)Context::kEvRing;
    const long long avail = std::min<long long>(fe->runs, keep);
    const long long take = std::min<long long>(avail, last_n < 1 ? avail : last_n);
    for (long long r = fe->runs - take; r < fe->runs; r++) {
        Context& c = fe->ctx[r % kPipe];
        const int slot = (int)((r / kPipe) % Context::kEvRing);
        HIPCHK(hipEventSynchronize(c.evFast1[slot]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c.evFast0[slot], c.evFast1[slot]));
        *sum_ms += ms; (*n_out)++;
    }
    return IVF_OK;
}

int ivf_frontend_pack_gather_block_of(ivf_frontend* fe, int age, uint8_t* d_block, size_t block_bytes, size_t* record_bytes,
                                      void* hip_stream)
{
    if (!fe || !record_bytes) return fail(IVF_E_INVALID, "null argument");
    const size_t nf = fe->ctx[0].hc.nfeatures;
    const size_t rec = 16 + nf * sizeof(ivf_keypoint) + nf * 32 + 2 * nf * sizeof(float);
    *record_bytes = rec;
    if (!d_block) return IVF_OK;
    if (age < 0 || age >= kPipe) return fail(IVF_E_INVALID, "age %d outside [0,%d): results stay valid for %d further runs", age, kPipe, kPipe - 1);
    if (fe->runs <= age) return fail(IVF_E_STATE, "no batch of age %d has run", age);
    const int k = (int)((fe->runs - 1 - age) % kPipe);
    const int np = fe->pairsOf[k];
    if (block_bytes < rec * np) return fail(IVF_E_CAPACITY, "gather block needs %zu bytes", rec * np);
    if (((size_t)d_block & 15) != 0) return fail(IVF_E_INVALID, "gather block must be 16-byte aligned");
    HIPCHK(hipSetDevice(fe->cfg.device_id));
    hipStream_t st = (hipStream_t)hip_stream;
    if (hip_stream == IVF_STREAM_OF_BATCH) st = fe->stream[k];        // in order behind the batch itself: nothing to wait for
    else HIPCHK(hipStreamWaitEvent(st, fe->evDone[k], 0));            // the batch ran on an internal stream
    launch_pack_gather(fe->ctx[k].b, (int)nf, np, d_block, rec, st);
    HIPCHK(hipGetLastError());
    return IVF_OK;
}

void* ivf_frontend_batch_stream(ivf_frontend* fe, int age)
{
    if (!fe || age < 0 || age >= kPipe || fe->runs <= age) return nullptr;
    return (void*)fe->stream[(fe->runs - 1 - age) % kPipe];
}

// A resident frame straight from a batch: keypoints, descriptors and uRight of one image go device -> device, the 64x48
// grid is built on the device; nothing but the 4-byte keypoint count crosses PCIe (the replays' small host mirror is fetched
// lazily by the first search).  `age` as in ivf_frontend_pack_gather_block_of; side 0 = left (with uRight), 1 = right.
int ivf_frame_create_from_frontend(ivf_frontend* fe, int age, int pair, int side, const ivf_bounds* bounds, ivf_frame** out)
{
    if (!out) return fail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (!fe || !bounds || side < 0 || side > 1) return fail(IVF_E_INVALID, "bad argument");
    if (!(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y)) return fail(IVF_E_INVALID, "empty image bounds");
    if (age < 0 || age >= kPipe || fe->runs <= age) return fail(IVF_E_STATE, "no batch of age %d is held", age);
    const int k = (int)((fe->runs - 1 - age) % kPipe);
    if (pair < 0 || pair >= fe->pairsOf[k]) return fail(IVF_E_INVALID, "pair %d outside the batch of %d", pair, fe->pairsOf[k]);
    const int dev = fe->cfg.device_id;
    HIPCHK(hipSetDevice(dev));
    const Buffers& b = fe->ctx[k].b;
    const size_t nf = fe->ctx[k].hc.nfeatures, img = (size_t)pair * 2 + side;
    ivf_frame* f = new ivf_frame();
    f->device = dev; f->bd = *bounds; f->hostKps = false; f->hostDesc = false;
    f->invW = (float)GC / (bounds->max_x - bounds->min_x); f->invH = (float)GR / (bounds->max_y - bounds->min_y);
    auto bail = [&](const char* what) { ivf_frame_destroy(f); return fail(IVF_E_NO_DEVICE, "%s failed for a frame from the front end", what); };
    // sized for the batch's capacity: the (pooled) arena is acquired before the keypoint count is known
    if (frame_alloc(f, (int)nf) != IVF_OK) { ivf_frame_destroy(f); return IVF_E_NO_DEVICE; }
    int n = 0;
    if (hipStreamWaitEvent(f->stream, fe->evDone[k], 0) != hipSuccess ||
        hipMemcpyAsync(&n, b.count + img, sizeof(int), hipMemcpyDeviceToHost, f->stream) != hipSuccess ||
        hipStreamSynchronize(f->stream) != hipSuccess) return bail("count read");
    f->n = n;
    if (n > 0) {
        if (hipMemcpyAsync(f->dKps, b.kps + img * nf, (size_t)n * sizeof(ivf_keypoint), hipMemcpyDeviceToDevice, f->stream) != hipSuccess ||
            hipMemcpyAsync(f->dDesc, b.desc + img * nf * 32, (size_t)n * 32, hipMemcpyDeviceToDevice, f->stream) != hipSuccess)
            return bail("device copy");
        if (side == 0) { if (hipMemcpyAsync(f->dUright, b.uright + (size_t)pair * nf, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, f->stream) != hipSuccess) return bail("device copy"); }
        else if (hipMemsetD32Async((hipDeviceptr_t)f->dUright, (int)0xbf800000, (size_t)n, f->stream) != hipSuccess) return bail("fill");      // -1.0f: no stereo
    }
    launch_grid_build(f->dKps, n, bounds->min_x, bounds->min_y, f->invW, f->invH, f->dStart, f->dIdx, f->stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(f->stream) != hipSuccess) return bail("grid build");
    *out = f;
    return IVF_OK;
}

int ivf_frontend_pack_gather_block(ivf_frontend* fe, uint8_t* d_block, size_t block_bytes, size_t* record_bytes, void* hip_stream)
{
    return ivf_frontend_pack_gather_block_of(fe, 0, d_block, block_bytes, record_bytes, hip_stream);
}

}  // extern "C"
