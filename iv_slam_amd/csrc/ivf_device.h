// ivf_device.h -- shared host/device structures of the gfx950 front end (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>      // getenv behind IVF_EXP_ENV (experiment builds)
#include "../../include/ivfront.h"

namespace ivf {

constexpr int kMaxLevels = IVF_MAX_LEVELS;
constexpr int kMaxCells = 2048;          // per level (LDS bookkeeping arrays in k_quota, 50 KB); the reference has ~N_level / 5 cells, i.e. ~10 000 features on one level
constexpr int kEdge = 19;                // EDGE_THRESHOLD (ORB/src/ORBextractor.cc:75)

// Geometry of one pyramid level.  All of it depends only on (params, image size), so the host
// computes it once (ORB/src/ORBextractor.cc:884-922) and the kernels read it as uniform data.
struct LevelGeom {
    int w, h, pitch;        // level size; row pitch in bytes (multiple of 64)
    int off;                // byte offset of the level inside one image's pyramid blob
    int nDesired;           // mnFeaturesPerLevel[level]
    int cols, rows, cellW, cellH, nCells, nfeaturesCell;
    int maxBX, maxBY;       // w-19, h-19
    int domH[2];            // FAST rows scanned per cell in non-last cell rows: [0] plain, [1] stale-hY (Appendix D-2)
    int domHLast;           // ... in the last cell row
    int winHLast;           // hY of the last cell row (cost-map pre-pass window)
    int cellBase;           // first cell of this level in per-image cell arrays
    int candBase, candCap;  // candidate scratch: element offset of the level, capacity per cell
    int kpBase;             // first keypoint slot of this level (prefix sum of nDesired)
    int scaledPatch;        // (int)(31*scale)
    int tileBase, tilesX, tilesY;   // FAST tile enumeration (scan region x>=16, y>=19)
    int btileBase, btilesX, btilesY; // blur tile enumeration (whole plane)
    int rtX, rtY;           // offsets into the packed cv::resize coefficient table (level >= 1)
    int valid;              // 0: level yields no keypoints
    float scale;            // mvScaleFactor[level]
    unsigned cellWMagic, cellHMagic;   // floor(2^32 / cellW) + 1: n / cellW = umulhi(n, magic), exact for n < 2^16 (cells are at most 4096 px)
};

struct Config {
    int nlevels, w, h;
    int nfeatures, iniTh, minTh, introspection;
    int pyrBytes;           // bytes of one image's pyramid blob
    int candTotal;          // candidate scratch elements per image
    int nCellsTotal;        // cells per image over all levels
    int nTiles;             // FAST tiles per image
    int nBlurTiles;         // blur tiles per image
    int varBlur, varRetain, varAtan;   // OpenCV-version switches (ivf_extractor_set_opencv_variant): 0 = OpenCV >= 3.4.2 / 4.x
    int maxCandCap;         // largest per-cell bound on strict 3x3 maxima over the levels (k_cell_select_huge slot size)
    int umax[16];
    float scale[kMaxLevels], invScale[kMaxLevels];
    int cellBases[kMaxLevels];                           // lv[l].cellBase of the VALID levels side by side (INT_MAX for an invalid level and past nlevels): one wide scalar load
    int tileBases[kMaxLevels], btileBases[kMaxLevels];   // lv[l].tileBase / btileBase side by side (INT_MAX past nlevels):
                                                         // a tile finds its level with one wide uniform load
    LevelGeom lv[kMaxLevels];
};

// cv::resize coefficient table entry: src index | coef0 << 16 | coef1 << 32 (11-bit fixed point, A-3)
typedef unsigned long long ResizeCoef;

#ifndef IVF_FAST_TH
#define IVF_FAST_TH 32        // 64 measured slower: 405 vs 348 us per 128 images (occupancy: 32 KB of LDS per workgroup, ragged level edges)
#endif
constexpr int kFastTW = 128, kFastTH = IVF_FAST_TH;    // FAST/NMS output tile (kFastTW + 2 <= 160, kFastTH + 2 <= 96: see k_fast_nms)
constexpr int kBlurTW = 128, kBlurTH = 32;    // blur output tile
constexpr int kTileCap = kFastTW * kFastTH / 4;   // at most one strict 3x3 maximum per 2x2 block
constexpr int kRowCap = 96;              // right keypoints listed per image row by k_stereo_rows (more: that row falls back to the full scan)
constexpr int kHugeSlots = 8;            // workgroups (= global scratch slots) of k_cell_select_huge
constexpr int kHugeListCap = 4096;       // cells with > 4096 survivors listed per launch

struct Buffers {            // device pointers of one batch context
    uint8_t* pyr;           // [nImg][pyrBytes]  un-blurred pyramid (level 0 = ingested input)
    uint8_t* qpyr;          // [nImg][pyrBytes]  cost-map pyramid (introspection) or nullptr
    uint8_t* blur;          // [nImg][pyrBytes]  7x7 sigma-2 blurred pyramid
    unsigned* tileList;     // [nImg][nTiles][kTileCap] unordered NMS survivors of each FAST tile: y<<20 | x<<8 | score
    int* tileCnt;           // [nImg][nTiles] survivors per tile
    int* cellCnt;           // [nImg][nCellsTotal][2] survivors per cell: {score >= minTh, score >= iniTh}
    unsigned long long* lvl;    // [nImg][candTotal] per level: kept keys (respbits<<32 | y<<16 | x) of all cells, (i,j) order
    int4* cellInfo;         // [nImg][nCellsTotal] {nTotal, nRetain, prefix, useMin} from k_quota
    int* lvlTotal;          // [nImg][kMaxLevels] length of each level list before the level-wide retainBest
    unsigned int* slotPos;  // [nImg][nfeatures] (y<<16|x) level coords
    float* slotResp;        // [nImg][nfeatures]
    int* lvlCount;          // [nImg][kMaxLevels]
    uint8_t* useCost;       // [nImg] bit 0 = cost pyramid gates this image's extraction, bit 1 = level 0 of the cost image feeds mvKeyQualScore
    ivf_keypoint* kps;      // [nImg][nfeatures]
    uint8_t* desc;          // [nImg][nfeatures][32]
    int* count;             // [nImg]
    float* quality;         // [nImg][nfeatures]  mvKeyQualScore (Frame.cc:130-143)
    float* uright;          // [nPairs][nfeatures]
    float* depth;           // [nPairs][nfeatures]
    int* sad;               // [nPairs][nfeatures]  best SAD distance or -1
    int* rowCnt;            // [nPairs][H]  right keypoints whose row band covers the row (k_stereo_rows)
    unsigned short* rowList;    // [nPairs][H][kRowCap]
    int* status;            // [1] device-side error flags, cleared by the host when read
    int* hugeCount;         // [3] cells with more than 4096 survivors in this launch (k_quota -> k_cell_select_huge); [1], [2]: lengths of the tier lists
    int* tierList;          // [2][nImg * nCellsTotal] cells with 257..1024 / 1025..4096 survivors: img * nCellsTotal + cell (k_quota -> k_cell_select_list)
    int* hugeList;          // [kHugeListCap] img * nCellsTotal + cell
    unsigned* hugeScratch;  // [kHugeSlots][6 * maxCandCap] dwords, or nullptr when no cell can exceed 4096 maxima
};

// arguments of the stereo matcher kernels: left/right data may live in one batch context
// (frontend: images interleaved L,R) or in two single-image contexts (two ivf_extractor handles)
struct StereoArgs {
    const uint8_t *pyrL, *pyrR; size_t pyrStride;          // bytes between consecutive pairs
    const ivf_keypoint *kpL, *kpR; const uint8_t *descL, *descR; const int *cntL, *cntR;
    size_t kpStride;        // elements (ivf_keypoint / 32-byte rows) between pairs
    int cntStride;
    float *uright, *depth; int* sad; int outStride;
    float bf, bb;
    int* rowCnt; unsigned short* rowList;                  // [nPairs][H], [nPairs][H][kRowCap]: right keypoints per image row, or null
};

// Environment switches.  The shipped library reads exactly the switches listed in INTEGRATION.md (documented, result-preserving:
// getenv is spelled out at those sites).  Every kernel-variant selector and tuning knob of the experiments behind DESIGN.md goes through
// IVF_EXP_ENV, which is getenv only in an experiment build (`make EXPERIMENT=1` -> libivfront_exp.so, loaded through IVFRONT_LIB by
// tools/ and by the kernel-variant tests) and a null constant in the product: the names are not even compiled in
// (tests/test_abi_cpu.py asserts the list of IVF_* strings in libivfront.so).
#ifdef IVF_EXPERIMENT
#define IVF_EXP_ENV(name) getenv(name)
#else
#define IVF_EXP_ENV(name) ((const char*)nullptr)
#endif

// every kernel launch of the library goes through hipLaunchKernelGGL: counted for bench.py's launches_per_step
void count_launch();
}  // namespace ivf
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...) do { ivf::count_launch(); hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__); } while (0)
namespace ivf {

// thread-local error message behind ivf_last_error() (ivf_api.hip)
int set_error(int code, const char* fmt, ...);

// launchers (ivf_kernels.hip)
void launch_stereo_args(const Config& hc, const Config* dc, const StereoArgs& A, int nPairs, hipStream_t s);
void launch_ingest(const Config& hc, const Config* dc, const Buffers& b, const uint8_t* src0, const uint8_t* src1,
                   size_t imageStride, int rowStride, int nImg, int nSides, uint8_t* dstBlob, hipStream_t s, int sideMask = 3);
void launch_ingest_color(const Config& hc, const Config* dc, const uint8_t* src, size_t imageStride, int rowStride, int code, int nImg, int nSides, int side,
                         uint8_t* dstBlob, hipStream_t s);
void launch_pyramid(const Config& hc, const Config* dc, const ResizeCoef* dTab, uint8_t* blob, uint8_t* qblob, const uint8_t* useCost,
                    int nImg, hipStream_t s);
void launch_fast(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s);
void launch_blur(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s, bool skipEmptyLevels);
void launch_select(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s);
void launch_describe(const Config& hc, const Config* dc, const Buffers& b, const uint8_t* cost0, size_t costStride,
                     int costPitch, int nImg, int nSides, hipStream_t s);
void launch_stereo(const Config& hc, const Config* dc, const Buffers& b, int nPairs, float bf, float bb, hipStream_t s);
void launch_test_retain_best(const float* dResp, int n, int nPoints, int* dOrder, hipStream_t s);
void launch_grid_build(const ivf_keypoint* kps, int n, float minX, float minY, float invW, float invH, int* start, int* idx, hipStream_t s);
void launch_grid_window(const ivf_keypoint* kps, const uint8_t* desc, const int* start, const int* idx, float minX, float minY,
                        float invW, float invH, int nq, const float* qu, const float* qv, const float* qr, const int* qminL,
                        const int* qmaxL, const uint8_t* qdesc, const uint8_t* qvalid, int cap, int* count, int* cand, hipStream_t s);
void launch_pack_gather(const Buffers& b, int nf, int nPairs, uint8_t* block, size_t recBytes, hipStream_t s);
void launch_hamming_pairs(const uint8_t* a, const uint8_t* b, const int* pairs, int n, int* dist, hipStream_t s);
void launch_distinct_median(const uint8_t* desc, int n, int* median, hipStream_t s);
void launch_bow_transform(const int* childStart, const int* child, const uint8_t* nodeDesc, const uint8_t* desc, int n, int nidLevel,
                          int* leaf, int* nodeAt, hipStream_t s);

}  // namespace ivf
