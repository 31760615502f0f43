// ivf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the IV-SLAM visual front end.
//
// Path (ORB/ = introspective_ORB_SLAM/):
//   k_ingest        input -> pitched level 0                       (ORBextractor::ComputePyramid :1298-1323, level 0)
//   k_pyr_down      level l <- bilinear(level l-1), cv::resize 8U   (:1311, :1341)
//   k_fast_nms      FAST-9/16 score + per-cell 3x3 NMS              (cv::FAST at :1045,:1051)
//   k_blur7         7x7 sigma-2 Gaussian, 8.8/16.16 fixed point     (:1276-1277)
//   k_quota / k_cell_select / k_level_select   threshold fallback, quotas, retainBest (ComputeKeyPointsOld :880-1213)
//   k_describe      IC_Angle + rBRIEF + output assembly             (:78-148, :1253-1294; Frame.cc:130-143)
//   k_stereo_match  row-band Hamming + 11x11 SAD + parabola         (Frame::ComputeStereoMatches Frame.cc:758-915)
//   k_stereo_gate   median gate                                     (Frame.cc:918-931)
//   k_hamming_pairs DescriptorDistance                              (ORBmatcher.cc:1700-1716)
//
// Everything here is integer/byte work bounded by HBM bandwidth; there is no GEMM shape in this
// part of the path, so no MFMA.  Compiled with -ffp-contract=off: float expressions must round
// exactly like the reference's un-fused CPU code.
#include "ivf_device.h"
#include <cstdlib>

namespace ivf {

#define DEVINL __device__ __forceinline__

static __device__ const int8_t __attribute__((aligned(16))) d_pattern[1024] = {
#include "../../include/ivf_pattern31.inc"
};

// The per-image flag byte through the scalar cache: gfx950 has no scalar byte load, so a plain `useCost [img]` is a vector load with an `s_waitcnt vmcnt(0)` in
// front of every scalar load that follows it at the head of a workgroup.  The flag array is allocated in whole dwords (Context::build).
DEVINL unsigned use_cost_of(const uint8_t* __restrict__ useCost, int img)
{
    const unsigned w = ((const unsigned*)useCost)[img >> 2];
    return (w >> (8 * (img & 3))) & 0xffu;
}

// sum over the wave without the LDS crossbar (r04; was six ds_bpermute round trips): row_shr 1/2/4/8 fold each row of 16 lanes into
// its lane 15, row_bcast15 / row_bcast31 fold the four rows into lane 63, which every lane then reads
DEVINL int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
// LDS operations of one wave execute in order: a wavefront-scope fence (compiler ordering) is all that one lane needs to
// read what another lane of the same wave wrote
DEVINL void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// XCD-aware work mapping for (tile, image) grids.  Blocks b and b + 8 share an XCD (round-robin dispatch), and every XCD has
// its own L2: with a plain (tile, image) grid the tiles of ONE image are spread over all eight L2s, so the halo rows /
// columns and the 128-byte lines that neighbouring tiles share are fetched from HBM once per XCD (measured: k_fast_nms 2.06x,
// k_blur7 3.1x its algorithmic bytes, profiles/r02_pmc_hbm_traffic.json).  Here image i is worked on by XCD i % 8 only.
// 1-D grid of ceil(nImg / 8) * 8 * nTiles blocks; returns false for the padding blocks.
DEVINL bool xcd_tile_image(int nTiles, int nImg, int& tile, int& img)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    img = (j / nTiles) * 8 + xcd; tile = j % nTiles;
    return img < nImg;
}
DEVINL unsigned wave_min_u32(unsigned v)
{
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:1
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:2
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:4
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;   // row_shr:8
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;   // row_bcast:15
    t = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ------------------------------------------------------------------------------------------------
// k_ingest: copy the caller's images (arbitrary row stride) into the pitched level-0 plane.
// One workgroup = kIngestRows rows of one image, 16 bytes per thread and step (r01: one dword per thread and a workgroup per
// kilobyte -- 192k workgroups for 128 pairs, 1.9 TB/s; the caller's rows are byte-aligned at best, which global loads tolerate);
// image i comes from src[i % nSides] + (i / nSides) * imageStride.
// ------------------------------------------------------------------------------------------------
constexpr int kIngestRows = 8;
__global__ __launch_bounds__(256) void k_ingest(const Config* __restrict__ cfg, const uint8_t* __restrict__ src0,
                                               const uint8_t* __restrict__ src1, size_t imageStride, int rowStride, int nSides,
                                               uint8_t* __restrict__ blob, const uint8_t* __restrict__ onlyFlagged, int sideMask)
{
    const LevelGeom& G = cfg->lv[0];
    const int img = blockIdx.y, y0 = blockIdx.x * kIngestRows;
    if (onlyFlagged && !(use_cost_of(onlyFlagged, img) & 3u)) return;            // a cost plane nothing reads (the right image of a stereo pair)
    if (!((sideMask >> (img % nSides)) & 1)) return;                // this side arrives in colour: k_ingest_color writes its plane
    const uint8_t* src = ((nSides == 2 && (img & 1)) ? src1 : src0) + (size_t)(img / nSides) * imageStride;
    uint8_t* dst = blob + (size_t)img * cfg->pyrBytes + G.off;
    const int q16 = G.pitch / 16;                                    // 16-byte pieces per pitched row (pitch % 64 == 0)
    const int rows = min(kIngestRows, G.h - y0);
    if (G.w >= 16) {
        // four pieces per thread requested before the first is stored (r04: one load -> store round trip per piece was the kernel's time).
        // r06: branch-free.  Every piece is the 16 bytes that END at min(x + 16, w) of its row, shifted down by the bytes that belong to the piece before
        // (0 for a whole piece, everything for a piece past the image): the row's last piece used to be a byte loop with a wait per byte -- up to 15
        // dependent round trips for one lane of EVERY wave at widths that are not a multiple of 16 (1242: 10), in front of the wave's other loads
        for (int i0 = threadIdx.x; i0 < rows * q16; i0 += 4 * 256) {
            unsigned __int128 t[4];
            size_t d[4];
            int nv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = min(i0 + 256 * u, rows * q16 - 1);
                const int r = i / q16, x = (i % q16) * 16;
                const int xe = min(x + 16, G.w);
                nv[u] = max(xe - x, 0);                              // bytes of this piece that are image: 16, 1..15 (the row's last), 0 (past it)
                d[u] = (size_t)(y0 + r) * G.pitch + x;
                __builtin_memcpy(&t[u], src + (size_t)(y0 + r) * rowStride + xe - 16, 16);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // (a slot past the workgroup's last piece was clamped to that piece: it stores the same bytes again -- a conditional store would let the
                // compiler sink that slot's load into the branch, behind a wait of its own)
                const unsigned __int128 v = (t[u] >> (4 * (16 - nv[u]))) >> (4 * (16 - nv[u]));   // two halves: a shift by all 128 bits (nv = 0) is not defined, a select is a branch
                __builtin_memcpy(dst + d[u], &v, 16);
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < rows * q16; i += 256) {            // narrower than one piece: bytes
        const int r = i / q16, x = (i % q16) * 16;
        const uint8_t* sp = src + (size_t)(y0 + r) * rowStride + x;
        unsigned w[4] = {0u, 0u, 0u, 0u};
        for (int k = 0; k < 16 && x + k < G.w; k++) w[k >> 2] |= (unsigned)sp[k] << (8 * (k & 3));
        *(uint4*)(dst + (size_t)(y0 + r) * G.pitch + x) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_ingest_color (r06): the same for a side that arrives as 8-bit 3-channel interleaved images -- the grey conversion in front of the extractor,
// Tracking::GrabImageStereo's cvtColor(mImGray, mImGray, CV_RGB2GRAY / CV_BGR2GRAY) (ORB/src/Tracking.cc:272-295, by mbRGB) fused into the ingest, so a
// PCIe-fed pipeline ships the left COLOUR image once (the FCN reads it too) instead of the colour image and a grey copy of it.
// cv::cvtColor 8UC3 -> 8UC1 is fixed point: OpenCV 4.x (R 9798 + G 19235 + B 3735 + 2^14) >> 15; OpenCV <= 3.x (R 4899 + G 9617 + B 1868 + 2^13) >> 14
// (code bit 2: IVF_COLOR_CV3).  code & 3: 1 = B,G,R byte order (CV_BGR2GRAY), 2 = R,G,B (CV_RGB2GRAY).
// One thread = 16 output pixels = 48 source bytes (three unaligned 16-byte loads).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest_color(const Config* __restrict__ cfg, const uint8_t* __restrict__ src, size_t imageStride, int rowStride,
                                                     int code, int nSides, int side, uint8_t* __restrict__ blob)
{
    const LevelGeom& G = cfg->lv[0];
    const int img = blockIdx.y * nSides + side, y0 = blockIdx.x * kIngestRows;
    const uint8_t* sI = src + (size_t)blockIdx.y * imageStride;
    uint8_t* dst = blob + (size_t)img * cfg->pyrBytes + G.off;
    const bool cv3 = (code & 4) != 0, rgb = (code & 3) == 2;
    const unsigned kR = cv3 ? 4899u : 9798u, kG = cv3 ? 9617u : 19235u, kB = cv3 ? 1868u : 3735u, half = cv3 ? 8192u : 16384u, sh = cv3 ? 14u : 15u;
    const unsigned k0 = rgb ? kR : kB, k2 = rgb ? kB : kR;
    const int q16 = G.pitch / 16;
    const int rows = min(kIngestRows, G.h - y0);
    for (int i = threadIdx.x; i < rows * q16; i += 256) {
        const int r = i / q16, x = (i % q16) * 16;
        const uint8_t* sp = sI + (size_t)(y0 + r) * rowStride + 3 * x;
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (x + 15 < G.w) {
            unsigned char px[48];
            __builtin_memcpy(px, sp, 48);
#pragma unroll
            for (int k = 0; k < 16; k++) w[k >> 2] |= ((px[3 * k] * k0 + px[3 * k + 1] * kG + px[3 * k + 2] * k2 + half) >> sh) << (8 * (k & 3));
        } else if (x < G.w) {
            // the row's last piece (1..15 pixels): every load at a clamped index, all in flight at once (a loop with a data-dependent bound waits per iteration)
            const int nv = G.w - x;
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const int kk = min(k, nv - 1);
                const unsigned g = (sp[3 * kk] * k0 + sp[3 * kk + 1] * kG + sp[3 * kk + 2] * k2 + half) >> sh;
                w[k >> 2] |= (k < nv ? g : 0u) << (8 * (k & 3));
            }
        }
        *(uint4*)(dst + (size_t)(y0 + r) * G.pitch + x) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pyr_down: cv::resize INTER_LINEAR 8UC1 (11-bit coefficients; horizontal pass in int,
// vertical pass ((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2).
// ONE launch per level serves the image pyramid (planes 0 .. nImg-1) and, when the batch carries cost maps, the cost pyramid of
// the images whose extraction it gates (planes nImg .. 2 nImg - 1; a plane whose useCost bit 0 is clear -- the right images of
// a stereo batch -- is skipped: nothing reads its upper levels).
// One workgroup = 256 x 32 output pixels.  The source rows it needs are staged in LDS by 16-byte loads, all of a thread's loads
// in flight together (r03 staged by dwords with an integer division per dword: 20 of its 49 VALU lane-instructions per output
// pixel).  The LDS tile is sized per level by the host (dynamic LDS: 14 KB at the 1.2 ratio instead of the 37 KB a ratio of 2
// needs) and the kernel keeps 40 registers, so eight workgroups share a CU: the kernel is bound by its dependent global -> LDS ->
// register round trips, not by issue (0.44 of the VALU rate) or HBM (2.8 TB/s), and occupancy is what hides them.  (Measured and
// dropped: workgroups walking four tiles with the next tile's loads in flight -- 36 more registers, half the occupancy, same time.)
// A thread produces 4 adjacent pixels of 8 CONSECUTIVE rows.  Per source row it reads three aligned dwords, funnels them into
// the 8-byte window its four columns draw from (v_alignbyte), picks each column's two source bytes as a u16 pair (v_perm with
// a per-column selector computed once) and gets p0*a0 + p1*a1 from one v_dot2_u32_u16; the row shared by two consecutive output
// rows is computed once; row indices and row coefficients are wave-uniform (scalar).  One dword store per row.  The per-column /
// per-row coefficients (fx = (float)((dx+0.5)*scale - 0.5), cvRound(f*2048) ...) come from a packed table the host builds once
// per geometry exactly as OpenCV's resize does.
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int kPyrTW = 256, kPyrTH = 32, kPyrRPT = kPyrTH / 4;             // rows per thread (4 waves stacked vertically)
constexpr int kPyrMaxP = 544, kPyrMaxR = 2 * kPyrTH + 4;                    // largest LDS tile: 68 rows x 544 B (level ratio <= 2)
// one 256 x 32 tile of level `level` of the plane at `base` (src = the workgroup's LDS tile of ldsRows x ldsPitch bytes)
DEVINL void pyr_tile(const Config* __restrict__ cfg, int level, const ResizeCoef* __restrict__ tab, uint8_t* __restrict__ base,
                     int dx0, int dy0, int ldsPitch, int ldsRows, uint8_t* src)
{
    const LevelGeom& D = cfg->lv[level];
    const LevelGeom& S = cfg->lv[level - 1];
    const uint8_t* SP = base + S.off;
    const ResizeCoef* tx = tab + D.rtX;
    const ResizeCoef* ty = tab + D.rtY;
    const int tid = threadIdx.x;
    // source window of this tile (table entries hold the clamped source index); rows start 64-byte aligned, the window at a
    // multiple of 16 bytes, and a pitched row always holds the whole last 16-byte piece
    const int dxl = min(dx0 + kPyrTW, D.w) - 1, dyl = min(dy0 + kPyrTH, D.h) - 1;
    const int wx0 = (int)(tx[dx0] & 0xffff) & ~15, wx1 = min((int)(tx[dxl] & 0xffff) + 1, S.w - 1);
    const int wy0 = (int)(ty[dy0] & 0xffff), wy1 = min((int)(ty[dyl] & 0xffff) + 1, S.h - 1);
    const int nq = (wx1 - wx0) / 16 + 1, nr = wy1 - wy0 + 1;
    const bool fits = nq * 16 <= ldsPitch && nr <= ldsRows && nq <= 64;
    if (fits) {
        // 8 rows x 32 pieces per pass; the 33rd / 34th piece of a row at ratios near 2 in a second, narrow sweep
        constexpr int kIt = (kPyrMaxR + 7) / 8;
        const int qc = tid & 31, r0 = tid >> 5;
        uint4 v[kIt];
        const uint8_t* gp = SP + (size_t)wy0 * S.pitch + wx0 + 16 * qc;
#pragma unroll
        for (int k = 0; k < kIt; k++) {
            const bool ok = qc < nq && r0 + 8 * k < nr;
            v[k] = *(const uint4*)(ok ? gp + (size_t)(r0 + 8 * k) * S.pitch : SP);
        }
        // r06: every loaded value passes through an empty asm HERE.  The stores below are conditional on the same test as the address select, so the compiler
        // moved each load into its store's branch: `if (ok) { load; s_waitcnt vmcnt(0); ds_write }` nine times in a row -- five or six SEQUENTIAL round trips per
        // thread at the 1.2 ratio, which is what "bound by its dependent round trips" (r04) really was.  The asm makes all of them live before the first store.
#pragma unroll
        for (int k = 0; k < kIt; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
#pragma unroll
        for (int k = 0; k < kIt; k++)
            if (qc < nq && r0 + 8 * k < nr) *(uint4*)(src + (r0 + 8 * k) * ldsPitch + 16 * qc) = v[k];
        if (nq > 32)
            for (int i = tid; i < (nq - 32) * nr; i += 256) {
                const int r = i / (nq - 32), q = 32 + i % (nq - 32);
                *(uint4*)(src + r * ldsPitch + 16 * q) = *(const uint4*)(SP + (size_t)(wy0 + r) * S.pitch + wx0 + 16 * q);
            }
    }
    __syncthreads();
    if (fits) {
        const int x4 = dx0 + (tid & 63) * 4;
        if (x4 >= D.pitch) return;
        // per column: the two source bytes as a zero-extended u16 pair picked out of the 8-byte window that starts at the first
        // column's source byte, and the coefficient pair
        unsigned sel[4], coef[4];
        int sxs[4], sx1s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const ResizeCoef cx = tx[min(x4 + k, D.w - 1)];
            sxs[k] = (int)(cx & 0xffff); sx1s[k] = min(sxs[k] + 1, S.w - 1);
            coef[k] = (unsigned)((cx >> 16) & 0xffff) | ((unsigned)((cx >> 32) & 0xffff) << 16);      // a0 | a1 << 16
        }
        // level-to-level ratios just above 2 (scaleFactor 2.0 with rounded level sizes) stretch the four columns over more than
        // 8 source bytes: such a thread draws columns 2 and 3 from a second window that starts at column 2's source byte
        const bool wide = sx1s[3] - sxs[0] > 7;
        const int base2 = wide ? sxs[2] : sxs[0];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int b = k < 2 ? sxs[0] : base2;
            sel[k] = (unsigned)(sxs[k] - b) | 0x0c00u | ((unsigned)(sx1s[k] - b) << 16) | 0x0c000000u;
        }
        const int wbyte = sxs[0] - wx0, wsh = wbyte & 3, wbyte2 = base2 - wx0, wsh2 = wbyte2 & 3;
        const unsigned* wrow = (const unsigned*)(src + (wbyte & ~3));
        const unsigned* wrow2 = (const unsigned*)(src + (wbyte2 & ~3));
        const int rowDw = ldsPitch / 4;
        // (h >> 4) of the four columns on staged row r
        auto hpass = [&](int r, unsigned (&h)[4]) {
            const unsigned* w = wrow + r * rowDw;
            const unsigned d0 = w[0], d1 = w[1], d2 = w[2];
            const unsigned lo = __builtin_amdgcn_alignbyte(d1, d0, wsh), hi = __builtin_amdgcn_alignbyte(d2, d1, wsh);
            unsigned lo2 = lo, hi2 = hi;
            if (wide) {
                const unsigned* w2 = wrow2 + r * rowDw;
                const unsigned e0 = w2[0], e1 = w2[1], e2 = w2[2];
                lo2 = __builtin_amdgcn_alignbyte(e1, e0, wsh2); hi2 = __builtin_amdgcn_alignbyte(e2, e1, wsh2);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(k < 2 ? hi : hi2, k < 2 ? lo : lo2, sel[k])),
                                              __builtin_bit_cast(u16x2, coef[k]), 0u, false) >> 4;
        };
        const unsigned colMask = x4 + 3 < D.w ? 0xffffffffu : (x4 >= D.w ? 0u : (0xffffffffu >> (8 * (x4 + 4 - D.w))));
        // a thread takes 8 CONSECUTIVE output rows: the lower source row of one output row is usually the upper one of the
        // next (scale 1.2), so its horizontal pass is reused.  The wave index, hence every row index and row coefficient
        // below, is uniform: scalar registers and scalar branches
        const int dyFirst = dy0 + kPyrRPT * __builtin_amdgcn_readfirstlane(tid >> 6);
        unsigned hA[4], hB[4];
        int rowA = -1, rowB = -1;                   // staged rows held in hA / hB
        uint8_t* outp = base + D.off + (size_t)dyFirst * D.pitch + x4;
        // r06: the eight row coefficients up front (scalar loads at clamped indices, one wait) instead of one scalar round trip at the head of every row
        unsigned cyl[kPyrRPT], cyh[kPyrRPT];
#pragma unroll
        for (int rr = 0; rr < kPyrRPT; rr++) {
            const ResizeCoef c = ty[min(dyFirst + rr, D.h - 1)];
            cyl[rr] = (unsigned)c; cyh[rr] = (unsigned)(c >> 32);
        }
#pragma unroll
        for (int rr = 0; rr < kPyrRPT; rr++) asm volatile("" : "+s"(cyl[rr]), "+s"(cyh[rr]));
#pragma unroll
        for (int rr = 0; rr < kPyrRPT; rr++) {
            const int dy = dyFirst + rr;
            if (dy >= D.h) break;
            const ResizeCoef cy = (ResizeCoef)cyl[rr] | ((ResizeCoef)cyh[rr] << 32);
            const int y0 = (int)(cy & 0xffff), y1 = min(y0 + 1, S.h - 1);
            const unsigned b0 = (unsigned)((cy >> 16) & 0xffff), b1 = (unsigned)((cy >> 32) & 0xffff);
            if (y0 - wy0 == rowB) {
#pragma unroll
                for (int k = 0; k < 4; k++) hA[k] = hB[k];
                rowA = rowB;
            } else if (y0 - wy0 != rowA) { hpass(y0 - wy0, hA); rowA = y0 - wy0; }
            if (y1 - wy0 == rowA) {
#pragma unroll
                for (int k = 0; k < 4; k++) hB[k] = hA[k];
                rowB = rowA;
            } else if (y1 - wy0 != rowB) { hpass(y1 - wy0, hB); rowB = y1 - wy0; }
            // every term is below 2^27 (coefficients <= 2048, h >> 4 <= 32640) and the result below 256: 24-bit multiplies, no masks
            unsigned out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                out |= (((__umul24(b0, hA[k]) >> 16) + (__umul24(b1, hB[k]) >> 16) + 2u) >> 2) << (8 * k);
            *(unsigned*)(outp + (size_t)rr * D.pitch) = out & colMask;
        }
        return;
    }
    // level ratios above 2 (source window larger than the LDS tile): per-pixel evaluation straight from global memory
    const int x4 = dx0 + (tid & 63) * 4;
    if (x4 >= D.pitch) return;
    // the 4 column coefficients of this thread are shared by its rows
    int sxk[4], sx1k[4], a0k[4], a1k[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = min(x4 + k, D.w - 1);
        const ResizeCoef cx = tx[dx];
        sxk[k] = (int)(cx & 0xffff); sx1k[k] = min(sxk[k] + 1, S.w - 1);
        a0k[k] = (int)((cx >> 16) & 0xffff); a1k[k] = (int)((cx >> 32) & 0xffff);
    }
    for (int rr = 0; rr < kPyrTH / 4; rr++) {
        const int dy = dy0 + (tid >> 6) + 4 * rr;
        if (dy >= D.h) break;
        const ResizeCoef cy = ty[dy];
        const int y0 = (int)(cy & 0xffff), y1 = min(y0 + 1, S.h - 1);
        const int b0 = (int)((cy >> 16) & 0xffff), b1 = (int)((cy >> 32) & 0xffff);
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned r = 0;
            if (x4 + k < D.w) {
                int p00, p01, p10, p11;
                const uint8_t* r0 = SP + (size_t)y0 * S.pitch;
                const uint8_t* r1 = SP + (size_t)y1 * S.pitch;
                p00 = r0[sxk[k]]; p01 = r0[sx1k[k]]; p10 = r1[sxk[k]]; p11 = r1[sx1k[k]];
                const int h0 = p00 * a0k[k] + p01 * a1k[k], h1 = p10 * a0k[k] + p11 * a1k[k];
                r = (unsigned)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
            }
            out |= r << (8 * k);
        }
        *(unsigned*)(base + D.off + (size_t)dy * D.pitch + x4) = out;
    }
}

__global__ __launch_bounds__(256) void k_pyr_down(const Config* __restrict__ cfg, int level, const ResizeCoef* __restrict__ tab,
                                                 uint8_t* __restrict__ blobI, uint8_t* __restrict__ blobQ,
                                                 const uint8_t* __restrict__ useCost, int nImg, int ldsPitch, int ldsRows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t src[];           // ldsRows x ldsPitch bytes (ldsPitch % 16 == 0)
    int img = blockIdx.z;
    uint8_t* blob = blobI;
    if (img >= nImg) { img -= nImg; blob = blobQ; if (!(use_cost_of(useCost, img) & 1u)) return; }
    pyr_tile(cfg, level, tab, blob + (size_t)img * cfg->pyrBytes, blockIdx.x * kPyrTW, blockIdx.y * kPyrTH, ldsPitch, ldsRows, src);
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 score of TWO horizontally adjacent pixels, packed 2 x i16 per register (v_pk_*_i16).
//   A = max over the 16 arcs of 9 contiguous ring pixels, both polarities, of the minimum signed
//   difference d = centre - ring; corner at t <=> A > t; score = A - 1 (OpenCV cornerScore<16>).
// W = the 8 bytes raw[row][x-3 .. x+4] of each of the 7 rows (x = left pixel); for ring offset dx the
// pair of ring bytes is (W[dx+3], W[dx+4]), pulled into the two i16 halves by one v_perm_b32.
// Quick reject first (every 9-arc holds one pixel of each opposite pair (k, k+8)); the sliding
// 9-window min/max runs on odd starts only: w8[j] = min d[j..j+7], arcs [j-1..j+7] and [j..j+8]
// give min(w8[j], max(d[j-1], d[j+8])) -- 47 packed ops per polarity instead of 79.
// ------------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
DEVINL s16x2 pkmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
DEVINL s16x2 pkmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
DEVINL s16x2 pair_at(unsigned lo, unsigned hi, int i)      // bytes W[i], W[i+1] of the window (hi:lo) -> 2 x i16
{
    return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | ((unsigned)(i + 1) << 16) | (unsigned)i));
}
// Cheap necessary condition on the 4 compass ring pixels only (rows -3, 0, +3): every 9-arc holds one pixel of each
// opposite pair, in particular of (0,8) = (0,+-3) and (4,12) = (+-3,0).  Passes ~10 % of the pairs (the full 8-pair
// test ~8 %) at a third of the cost.  loT/hiT = row -3, loC/hiC = centre row, loB/hiB = row +3 windows.
// the compass test on the ring pixels themselves: with d = v - p,  min(max(d0, d8), max(d4, d12)) > t  <=>  max(min(p0, p8), min(p4, p12)) < v - t
// and  max(min(d0, d8), min(d4, d12)) < -t  <=>  min(max(p0, p8), max(p4, p12)) > v + t: the four differences are never formed
DEVINL bool compass_test(s16x2 v, s16x2 p0, s16x2 p8, s16x2 p4, s16x2 p12, int t)
{
    const s16x2 T = {(short)t, (short)t};
    const s16x2 X = pkmax(pkmin(p0, p8), pkmin(p4, p12)), Y = pkmin(pkmax(p0, p8), pkmax(p4, p12));
    return ((__builtin_bit_cast(unsigned, X - (v - T)) | __builtin_bit_cast(unsigned, (v + T) - Y)) & 0x80008000u) != 0u;
}
DEVINL bool fast_precheck_pair(unsigned loT, unsigned hiT, unsigned loC, unsigned hiC, unsigned loB, unsigned hiB, int t)
{
    const s16x2 v = pair_at(loC, hiC, 3);
    return compass_test(v, pair_at(loB, hiB, 3), pair_at(loT, hiT, 3), pair_at(loC, hiC, 6), pair_at(loC, hiC, 0), t);
}
// the same test for the pixel pair at bytes 5, 6 of the 12-byte span (m0, m1, m2): x-3 = bytes 2, 3; x+3 = bytes 8, 9 = bytes 4, 5 of (m2:m1)
DEVINL bool fast_precheck_pair_b(unsigned t0, unsigned t1, unsigned m0, unsigned m1, unsigned m2, unsigned b0, unsigned b1, int t)
{
    const s16x2 v = pair_at(m0, m1, 5);
    return compass_test(v, pair_at(b0, b1, 5), pair_at(t0, t1, 5), pair_at(m1, m2, 4), pair_at(m0, m1, 2), t);
}
// rows: 7 windows (dy = -3..3), each as (lo, hi) dwords.  Returns the two scores packed (left | right << 16).
DEVINL unsigned fast_score_pair(const unsigned (&lo)[7], const unsigned (&hi)[7], int t)
{
    // r04: everything on the ring pixels p themselves instead of on the differences d = v - p (16 subtractions fewer per pair):
    //   8-pair test:  min_k max(d_k, d_k+8) > t  <=>  max_k min(p_k, p_k+8) < v - t;   max_k min(d_k, d_k+8) < -t  <=>  min_k max(p_k, p_k+8) > v + t
    //   one polarity per pixel (a 9-arc holds BOTH pixels of one opposite pair, so where every pair holds a ring pixel brighter than
    //   v + t only the "brighter ring" polarity can make a corner, elsewhere only the "darker ring" one, and the other polarity's arc
    //   value is <= t there): with sg = -1 / +1 and q = sg p, the arc minimum of e = sg d is  sg v - (maximum of q over the arc), so
    //   A = sg v - min over the 16 arcs of the arc maximum of q: the same sliding network with min and max swapped.
    const s16x2 v = pair_at(lo[3], hi[3], 3);
    s16x2 p[16];
    // ring order (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    p[0] = pair_at(lo[6], hi[6], 3);   p[1] = pair_at(lo[6], hi[6], 4);   p[2] = pair_at(lo[5], hi[5], 5);
    p[3] = pair_at(lo[4], hi[4], 6);   p[4] = pair_at(lo[3], hi[3], 6);   p[5] = pair_at(lo[2], hi[2], 6);
    p[6] = pair_at(lo[1], hi[1], 5);   p[7] = pair_at(lo[0], hi[0], 4);   p[8] = pair_at(lo[0], hi[0], 3);
    p[9] = pair_at(lo[0], hi[0], 2);   p[10] = pair_at(lo[1], hi[1], 1);  p[11] = pair_at(lo[2], hi[2], 0);
    p[12] = pair_at(lo[3], hi[3], 0);  p[13] = pair_at(lo[4], hi[4], 0);  p[14] = pair_at(lo[5], hi[5], 1);
    p[15] = pair_at(lo[6], hi[6], 2);
    s16x2 X = pkmin(p[0], p[8]), Y = pkmax(p[0], p[8]);
#pragma unroll
    for (int k = 1; k < 8; k++) { X = pkmax(X, pkmin(p[k], p[k + 8])); Y = pkmin(Y, pkmax(p[k], p[k + 8])); }
    const s16x2 T = {(short)t, (short)t};
    const s16x2 u = X - (v - T), w = (v + T) - Y;                                // negative where the darker / brighter 8-pair test passes
    if (((__builtin_bit_cast(unsigned, u) | __builtin_bit_cast(unsigned, w)) & 0x80008000u) == 0u) return 0u;
    const s16x2 sg = (w >> 15) | (s16x2){1, 1};
    s16x2 q[16];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = p[k] * sg;
    s16x2 mx[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { const int j = 2 * i + 1; mx[i] = pkmax(q[j], q[(j + 1) & 15]); }
    s16x2 m4[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m4[i] = pkmax(mx[i], mx[(i + 1) & 7]);
    s16x2 wmin;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = 2 * i + 1;
        const s16x2 w8 = pkmax(m4[i], m4[(i + 2) & 7]);                          // q[j..j+7]
        const s16x2 a = pkmax(w8, pkmin(q[(j + 15) & 15], q[(j + 8) & 15]));     // arcs [j-1..j+7] and [j..j+8]
        wmin = i ? pkmin(wmin, a) : a;
    }
    const s16x2 amax = v * sg - wmin;
    const int A0 = (int)amax.x, A1 = (int)amax.y;
    const unsigned s0 = A0 > t ? (unsigned)(A0 - 1) : 0u, s1 = A1 > t ? (unsigned)(A1 - 1) : 0u;
    return s0 | (s1 << 16);
}

// ------------------------------------------------------------------------------------------------
// k_fast_nms: one workgroup = one 128x32 output tile (kFastTW x kFastTH) of one level of one image.
//   1. stage the (128+8)x(32+8) raw tile in LDS with aligned dword loads (tile origin x = 16 + 128*tx);
//      ~160 threads also classify the tile's 130 columns / 34 rows against the level's cell grid once
//      (which cell, inside a FAST detection domain or not, neighbours in the same cell or not)
//   2. FAST score for the 130x34 region (tile + 1-px NMS halo), two pixels per thread-step
//   3. 3x3 strict NMS against neighbours of the SAME cell (cv::FAST runs per cell sub-image, so
//      neighbours in another cell count as 0); every survivor (score >= minTh) goes to the tile's list as
//      (y<<20 | x<<8 | score) and is counted per cell (total and >= iniTh).  No score map goes to HBM;
//      k_cell_select collects a cell's survivors from the tiles it overlaps and restores cv::FAST's
//      row-major order from the packed positions.
// Since score >= t <=> corner at t, one pass serves both thresholds.
// ------------------------------------------------------------------------------------------------
constexpr int kRawP = kFastTW + 8;          // 136 bytes per raw row (34 dwords)
constexpr int kLocalCells = 32;             // cells a tile may touch on the aggregated path: 8 cell rows x 4 cell cols
constexpr int kScW = kFastTW + 2, kScH = kFastTH + 2, kScP = kFastTW + 4;   // score region 130 x 34, pitch 132
// per column / row of the score region: bit0 valid, bit1 previous neighbour in the same cell, bit2 next neighbour
// in the same cell, bits 8.. = cell column / row index
__global__ __launch_bounds__(256) void k_fast_nms(const Config* __restrict__ cfg, const uint8_t* __restrict__ pyr,
                                                 const uint8_t* __restrict__ useCost, unsigned* __restrict__ tileList,
                                                 int* __restrict__ tileCnt, int* __restrict__ cellCnt, int nImg)
{
    __shared__ __attribute__((aligned(16))) unsigned raw[(kFastTH + 8) * (kRawP / 4) + 4];   // +4: the funnel read touches one dword past a window
    __shared__ __attribute__((aligned(16))) uint8_t sc[(kScH * kScP + 15) / 16 * 16];
    __shared__ unsigned colInfo[kScW + 2], rowInfo[kScH + 2];
    __shared__ int s_cnt[kLocalCells], s_ini[kLocalCells], s_n;
    // s_queue: pairs that pass the quick test (sy << 8 | sx), dead after pass B; s_list (survivors, at most one strict
    // 3x3 maximum per 2x2 block) reuses its storage
    // (r04: one region per wave, filled through a wave-uniform running count -- no LDS atomic and no shuffle per step)
    constexpr int kQuadsA = (kScW + 3) / 4, kRG = 256 / kQuadsA, kRPT = (kScH + kRG - 1) / kRG;   // pass A: thread = quad column x kRPT consecutive rows
    constexpr int kQRegion = 64 * kRPT * 2;
    static_assert(kRG * kRPT >= kScH && kRG >= 1, "pass A covers the score region");
    __shared__ unsigned s_qmem[4 * kQRegion / 2 > kFastTW * kFastTH / 4 ? 4 * kQRegion / 2 : kFastTW * kFastTH / 4];
    __shared__ int s_nqw[4];
    unsigned short* const s_queue = (unsigned short*)s_qmem;
    unsigned* const s_list = s_qmem;
    constexpr int kCandCap = 32 * kFastTH;
    static_assert(kScW <= 160 && 160 + kScH <= 256, "column classes: threads 0..kScW-1, row classes: threads 160..160+kScH-1");
    __shared__ unsigned short s_cand[kCandCap];                       // tile pixels with a non-zero score (sy << 8 | sx)
    __shared__ int s_nc;
    int img, bx;
    if (!xcd_tile_image(cfg->nTiles, nImg, bx, img)) return;
    const int tid = threadIdx.x;
    // the prologue is latency: every load below is issued before anything waits on one.  Level lookup: all tile bases
    // in one wide load (INT_MAX past nlevels)
    const unsigned useC = use_cost_of(useCost, img);
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) level += bx >= cfg->tileBases[l] ? 1 : 0;   // bases ascend
    const LevelGeom G = cfg->lv[level];                         // one uniform copy: two wide scalar loads
    const int t = bx - G.tileBase;
    if (!G.valid || t >= G.tilesX * G.tilesY) { if (tid == 0) tileCnt[(size_t)img * cfg->nTiles + bx] = 0; return; }
    const int tx = t % G.tilesX, ty = t / G.tilesX;
    const int x0 = 16 + tx * kFastTW, y0 = kEdge + ty * kFastTH;
    const uint8_t* src = pyr + (size_t)img * cfg->pyrBytes + G.off;

    // 1. raw tile: rows y0-4 .. y0+TH+3, cols x0-4 .. x0+TW+3 (x0-4 is a multiple of 4).  Branch-free clamped
    // addresses so the loads of a thread are all in flight together; out-of-plane dwords are zeroed afterwards.
    // thread = raw column dword (tid % 34) x rows (tid / 34) + 7 k: one division per thread instead of one per dword
    constexpr int kRawQ = kRawP / 4, kRawRPP = 256 / kRawQ, kRawIt = (kFastTH + 8 + kRawRPP - 1) / kRawRPP;
    const int rawQ = tid % kRawQ, rawR = tid / kRawQ;
    unsigned rv_[kRawIt];
    {
        // buffer loads (r04): the plane is the buffer, so rows below the plane read as zero by the hardware's range check and a thread's
        // address is one 32-bit offset advanced by a constant -- no clamps, no 64-bit address arithmetic, no select at the store
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, G.pitch * G.h, 0x00020000);
        const int gx = x0 - 4 + 4 * rawQ;
        unsigned off = (rawR < kRawRPP && gx < G.pitch) ? (unsigned)((y0 - 4 + rawR) * G.pitch + gx) : 0x7fffffffu;   // gy >= 0: y0 >= kEdge
        const unsigned step = (unsigned)(kRawRPP * G.pitch);
#pragma unroll
        for (int k = 0; k < kRawIt; k++) {
            rv_[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
            off += step;                        // an invalid thread stays far beyond the plane
        }
    }
    // the score plane starts as zeros (pass B writes the scored pairs only): wide stores while the loads are in flight
    for (int i = tid; i < (int)(sizeof(sc) / 16); i += 256) ((uint4*)sc)[i] = make_uint4(0u, 0u, 0u, 0u);
    const int mode = (cfg->introspection && (useC & 1u)) ? 1 : 0;
    const int domHm = mode ? G.domH[1] : G.domH[0];
    if (tid < kScW) {                               // column classes: x = x0-1+tid
        const int x = x0 - 1 + tid;
        unsigned f = 0;
        if (x >= kEdge && x < G.maxBX) {
            const int j = G.cellW == 1 ? x - kEdge : (int)__umulhi((unsigned)(x - kEdge), G.cellWMagic);      // (x - kEdge) / cellW
            const int cx0 = kEdge + j * G.cellW, cx1 = (j == G.cols - 1) ? G.maxBX : cx0 + G.cellW;
            if (x < cx1) f = 1u | ((x - 1 >= cx0) ? 2u : 0u) | ((x + 1 < cx1) ? 4u : 0u) | ((unsigned)j << 8);
        }
        colInfo[tid] = f;
        if (tid < 2) colInfo[kScW + tid] = 0;
    } else if (tid >= 160 && tid < 160 + kScH) {    // row classes: y = y0-1+(tid-160)
        const int y = y0 - 1 + (tid - 160);
        unsigned f = 0;
        if (y >= kEdge && y < G.maxBY) {
            int i = G.cellH == 1 ? y - kEdge : (int)__umulhi((unsigned)(y - kEdge), G.cellHMagic);            // (y - kEdge) / cellH
            if (i > G.rows - 1) i = G.rows - 1;
            const int cy0 = kEdge + i * G.cellH, cy1 = cy0 + ((i == G.rows - 1) ? G.domHLast : domHm);
            if (y < cy1) f = 1u | ((y - 1 >= cy0) ? 2u : 0u) | ((y + 1 < cy1) ? 4u : 0u) | ((unsigned)i << 8);
        }
        rowInfo[tid - 160] = f;
    }
#pragma unroll
    for (int k = 0; k < kRawIt; k++)
        if (rawR < kRawRPP && rawR + kRawRPP * k < kFastTH + 8) raw[(rawR + kRawRPP * k) * kRawQ + rawQ] = rv_[k];
    __syncthreads();
    // 2. scores, four pixels (two packed pairs) per step sharing the three aligned dwords of each of the 7 rows:
    //    score columns sx..sx+3 (sx % 4 == 0); the window of pair A (sx, sx+1) is bytes 0..7 of the 12-byte span
    //    starting at raw byte sx, the window of pair B (sx+2, sx+3) is bytes 2..9
    const int minTh = cfg->minTh;
    constexpr int kQuads = (kScW + 3) / 4;
    // pass A (all pixels): compass pre-check only (3 of the 7 rows).  ~10 % of the pairs pass, but in most waves at
    // least one lane does, so nothing expensive runs here: passing pairs are queued in LDS and scored densely in pass B.
    if (tid == 0) s_nc = 0;
    {
        // thread = one quad column x kRPT consecutive rows: the column classes are read once, the 11 raw rows a thread's 5 steps touch are
        // requested up front (each is the top row of one step, the centre row of another, the bottom row of a third)
        const int q = tid % kQuads, rg = tid / kQuads, sx = 4 * q, sy0 = rg * kRPT;
        const bool thr = rg < kRG;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        unsigned short* const myq = s_queue + wv * kQRegion;
        const bool cA = thr && ((colInfo[sx] | colInfo[sx + 1]) & 1u), cB = thr && ((colInfo[sx + 2] | colInfo[sx + 3]) & 1u);
        unsigned w0[kRPT + 6], w1[kRPT + 6], w2[kRPT + 6];
#pragma unroll
        for (int j = 0; j < kRPT + 6; j++) {
            const unsigned* base = raw + min(sy0 + j, kFastTH + 7) * (kRawP / 4) + q;
            w0[j] = base[0]; w1[j] = base[1];
            w2[j] = (j >= 3 && j < kRPT + 3) ? base[2] : 0u;
        }
        unsigned rvv[kRPT];
#pragma unroll
        for (int k = 0; k < kRPT; k++) rvv[k] = rowInfo[min(sy0 + k, kScH - 1)];
        int nw = 0;                             // wave-uniform count of this wave's region
        const int lane = tid & 63;
#pragma unroll
        for (int k = 0; k < kRPT; k++) {
            const int sy = sy0 + k;
            const bool in = thr && sy < kScH, rowOk = in && (rvv[k] & 1u);
            bool pA = false, pB = false;
            if (rowOk && cA) pA = fast_precheck_pair(w0[k], w1[k], w0[k + 3], w1[k + 3], w0[k + 6], w1[k + 6], minTh);
            // pair B sits two bytes further: its five operands are picked straight out of the same dwords (only x+3 reaches the third one)
            if (rowOk && cB) pB = fast_precheck_pair_b(w0[k], w1[k], w0[k + 3], w1[k + 3], w2[k + 3], w0[k + 6], w1[k + 6], minTh);
            const unsigned long long mA = __ballot(pA), mB = __ballot(pB);
            const int nA = __popcll(mA);
            if (pA) myq[nw + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u))] = (unsigned short)((sy << 8) | sx);
            if (pB) myq[nw + nA + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u))] = (unsigned short)((sy << 8) | (sx + 2));
            nw += nA + __popcll(mB);
        }
        if (lane == 0) s_nqw[wv] = nw;
    }
    __syncthreads();
    // pass B (queued pairs only): full score
    const int nq0 = s_nqw[0], nq1 = nq0 + s_nqw[1], nq2 = nq1 + s_nqw[2], nq = nq2 + s_nqw[3];
    for (int i = tid; i < nq; i += 256) {
        const int qi = i < nq0 ? i : (i < nq1 ? kQRegion + i - nq0 : (i < nq2 ? 2 * kQRegion + i - nq1 : 3 * kQRegion + i - nq2));
        const unsigned qe = s_queue[qi];
        const int sy = qe >> 8, sx = qe & 0xff;
        unsigned lo[7], hi[7];
        const int sh = sx & 2;
        const unsigned* base = raw + (sy * kRawP + (sx & ~3)) / 4;
#pragma unroll
        for (int r = 0; r < 7; r++) {
            // rows +-3 are read at window bytes 2 .. 5 only (ring dx = -1, 0, 1): their third dword is never looked at
            const unsigned w0 = base[r * (kRawP / 4)], w1 = base[r * (kRawP / 4) + 1], w2 = (r == 0 || r == 6) ? w1 : base[r * (kRawP / 4) + 2];
            lo[r] = __builtin_amdgcn_alignbyte(w1, w0, sh);
            hi[r] = __builtin_amdgcn_alignbyte(w2, w1, sh);
        }
        unsigned two = fast_score_pair(lo, hi, minTh);
        if (!(colInfo[sx] & 1u)) two &= 0xffff0000u;
        if (!(colInfo[sx + 1] & 1u)) two &= 0x0000ffffu;
        const unsigned packed = (two & 0xffu) | ((two >> 8) & 0xff00u);
        *(unsigned short*)(sc + sy * kScP + sx) = (unsigned short)packed;
        // scored pixels of the tile proper (not the 1-px halo) go to the NMS list, one LDS atomic per wave step
        const bool rowIn = sy >= 1 && sy <= kFastTH;
        const bool cA = rowIn && (packed & 0xffu) && sx >= 1 && sx <= kFastTW;
        const bool cB = rowIn && (packed >> 8) && sx + 1 <= kFastTW;
        const unsigned long long mA = __ballot(cA), mB = __ballot(cB);
        if (mA | mB) {
            const int lane = tid & 63, lead = __ffsll((long long)(mA | mB)) - 1;
            const int nA = __popcll(mA);
            int qb = 0;
            if (lane == lead) qb = atomicAdd(&s_nc, nA + __popcll(mB));
            qb = __shfl(qb, lead, 64);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int iA = qb + __popcll(mA & lt), iB = qb + nA + __popcll(mB & lt);
            if (cA && iA < kCandCap) s_cand[iA] = (unsigned short)((sy << 8) | sx);
            if (cB && iB < kCandCap) s_cand[iB] = (unsigned short)((sy << 8) | (sx + 1));
        }
    }
    __syncthreads();
    // 3. NMS, then publish.  Survivors are gathered in an LDS list and leave as ONE coalesced copy into the tile's own
    // slot of `tileList` (arbitrary order, count in `tileCnt`): no slot reservation, no returning atomics.  Only the
    // per-cell counters (needed by k_quota) use global atomics, non-returning, one per cell the tile touches.
    const int iniTh = cfg->iniTh;
    if (tid < kLocalCells) { s_cnt[tid] = 0; s_ini[tid] = 0; }
    if (tid == 0) s_n = 0;
    // local cell id = (cell row - first cell row of the tile) * 4 + (cell col - first cell col of the tile)
    const int firstCol = (int)(colInfo[x0 - 1 < kEdge ? kEdge - (x0 - 1) : 1] >> 8);
    int fr = 0;
    for (int q = 1; q <= kFastTH; q++) if (rowInfo[q] & 1u) { fr = (int)(rowInfo[q] >> 8); break; }
    const int firstRow = fr;
    int* cnt = cellCnt + (size_t)img * cfg->nCellsTotal * 2;
    __syncthreads();
    auto nms_one = [&](int sy, int sx, int s) {
        const uint8_t* c = sc + sy * kScP + sx;
        const unsigned ci = colInfo[sx], ri = rowInfo[sy];
        if (!(ci & ri & 1u)) return;
        // strict maximum over the 8 neighbours, a neighbour in another cell counting as 0: all-ones / zero masks from the class bits and
        // one max3 per row (sc has a zero column / row around the region, so every address is inside the array)
        const int mL = -(int)((ci >> 1) & 1u), mR = -(int)((ci >> 2) & 1u), mU = -(int)((ri >> 1) & 1u), mD = -(int)((ri >> 2) & 1u);
        const int up = max(max((int)c[-kScP - 1] & mL, (int)c[-kScP]), (int)c[-kScP + 1] & mR) & mU;
        const int dn = max(max((int)c[kScP - 1] & mL, (int)c[kScP]), (int)c[kScP + 1] & mR) & mD;
        if (!(s > max(max(up, dn), max((int)c[-1] & mL, (int)c[1] & mR)))) return;
        const int crow = (int)(ri >> 8), ccol = (int)(ci >> 8);
        const int lr = crow - firstRow, lc = ccol - firstCol;
        // kTileCap (one strict maximum per 2x2 block) is not a strict bound: maxima on both sides of a CELL border do not see each
        // other.  A tile that exceeds it keeps its true count in tileCnt, so the selection kernels' consistency check (survivors
        // gathered != survivors counted) reports the batch instead of LDS being overrun here.
        const int pos = atomicAdd(&s_n, 1);
        if (pos < kTileCap) s_list[pos] = ((unsigned)(y0 + sy - 1) << 20) | ((unsigned)(x0 + sx - 1) << 8) | (unsigned)s;
        if (lr >= 0 && lr < kLocalCells / 4 && lc >= 0 && lc < 4) {
            atomicAdd(&s_cnt[lr * 4 + lc], 1);
            if (s >= iniTh) atomicAdd(&s_ini[lr * 4 + lc], 1);
        } else {                                                    // tile spans too many cells: count directly
            const int gc = G.cellBase + crow * G.cols + ccol;
            atomicAdd(&cnt[2 * gc], 1);
            if (s >= iniTh) atomicAdd(&cnt[2 * gc + 1], 1);
        }
    };
    const int nc = s_nc;
    if (nc <= kCandCap) {
        // the usual case: dense over the scored pixels pass B listed, one per lane
        for (int i = tid; i < nc; i += 256) {
            const int sy = s_cand[i] >> 8, sx = s_cand[i] & 0xff;
            nms_one(sy, sx, sc[sy * kScP + sx]);
        }
    } else {
        // corner-dense tile (the list overflowed): scan the score plane, four pixels per 32-bit LDS read
        // (sc rows start at score column 0 = x0-1; the aligned dword at byte 4q holds score columns 4q .. 4q+3)
        for (int i = tid; i < kFastTH * (kFastTW / 4 + 1); i += 256) {
            const int oy = i / (kFastTW / 4 + 1), q = i % (kFastTW / 4 + 1);
            const unsigned four = *(const unsigned*)(sc + (oy + 1) * kScP + 4 * q);
            if (four == 0) continue;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sx = 4 * q + k;
                const int s = (four >> (8 * k)) & 0xff;
                if (s == 0 || sx < 1 || sx > kFastTW) continue;
                nms_one(oy + 1, sx, s);
            }
        }
    }
    __syncthreads();
    if (tid < kLocalCells && s_cnt[tid] > 0) {
        const int gc = G.cellBase + (firstRow + tid / 4) * G.cols + firstCol + (tid & 3);
        atomicAdd(&cnt[2 * gc], s_cnt[tid]);
        if (s_ini[tid]) atomicAdd(&cnt[2 * gc + 1], s_ini[tid]);
    }
    const int n = s_n;
    unsigned* out = tileList + ((size_t)img * cfg->nTiles + bx) * kTileCap;
    for (int i = tid; i < min(n, kTileCap); i += 256) out[i] = s_list[i];
    if (tid == 0) tileCnt[(size_t)img * cfg->nTiles + bx] = n;
}

// ------------------------------------------------------------------------------------------------
// k_blur7: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in OpenCV's 8-bit fixed point:
// kernel [18,34,48,56,48,34,18]/256 (A-4; [18,34,49,55,...] under the <= 3.4.1 switch), exact horizontal pass (u16),
// vertical pass (+32768)>>16, saturated.  One workgroup = one 128x32 output tile:
//   1. raw rows y0-3 .. y0+34 as aligned dwords (x0-4 .. x0+131), reflect-101 at the plane's borders
//   2. horizontal pass on BYTES with v_dot4_u32_u8: pixel x = dot4(bytes x-3..x, k0..k3) + dot4(bytes x+1..x+4, k4..k6,0); the
//      two 4-byte windows of each of a lane's 4 pixels come out of three dwords with v_alignbyte -- 3.5 instructions per
//      pixel instead of 12 byte extractions + 7 multiply-adds; results to LDS as u16
//   3. vertical pass, a lane = 4 columns x 4 output rows: 10 row reads (b64 = 4 x u16) for 4 rows, v_mad_u32_u16 with
//      op_sel picks the high / low u16 without unpacking (7 per pixel), bytes 2 of the four sums are the outputs.
// r01's version spent 46 VALU lane-instructions per pixel (SQ counters: VALU-bound at 1.6 TB/s effective); this one ~19.
// ------------------------------------------------------------------------------------------------
DEVINL int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}
DEVINL unsigned mad_u16_lo(unsigned a, unsigned k, unsigned c)
{ unsigned r; asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
DEVINL unsigned mad_u16_hi(unsigned a, unsigned k, unsigned c)
{ unsigned r; asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
__global__ __launch_bounds__(256) void k_blur7(const Config* __restrict__ cfg, const uint8_t* __restrict__ pyr,
                                              const int* __restrict__ lvlCount, uint8_t* __restrict__ blur, int nImg)
{
    constexpr int RQ = (kBlurTW + 8) / 4, RH = kBlurTH + 6, HQ = kBlurTW / 4;     // 34 dwords x 38 rows raw; 32 quads per row
    static_assert(RH % 2 == 0 && HQ == 32 && kBlurTH == 32, "row pairs, 32 quads x 8 row groups");
    __shared__ unsigned raw[RH * RQ];
    __shared__ __attribute__((aligned(16))) uint4 hp[(RH / 2) * HQ];              // per quad: 4 x (row 2p | row 2p+1 << 16), u16 each
    int img, bx;
    if (!xcd_tile_image(cfg->nBlurTiles, nImg, bx, img)) return;
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) level += bx >= cfg->btileBases[l] ? 1 : 0;      // one wide uniform load (INT_MAX past nlevels)
    const LevelGeom& G = cfg->lv[level];
    const int tilesX = G.btilesX, tilesY = G.btilesY;
    const int t = bx - G.btileBase;
    if (t >= tilesX * tilesY) return;
    if (lvlCount && lvlCount[img * kMaxLevels + level] == 0) return;   // :1270 levels without keypoints are skipped
    const int x0 = (t % tilesX) * kBlurTW, y0 = (t / tilesX) * kBlurTH;
    const uint8_t* src = pyr + (size_t)img * cfg->pyrBytes + G.off;
    const int tid = threadIdx.x;
    const int gw = G.w, gh = G.h, pitch = G.pitch;
    if (gw >= 8 && gh >= 8) {
        // the usual case: a tap leaves the plane by at most 3 (rows) / 7 (the dword past the right edge) pixels, so ONE
        // reflection is enough -- closed form, no loops; all of a thread's loads are issued before the first LDS store.
        // Thread = raw column dword (tid % 34) x rows (tid / 34) + 7 k: one division per thread, not one per dword.
        constexpr int RPP = 256 / RQ, IT = (RH + RPP - 1) / RPP;     // 7 rows per pass, 6 passes
        const int rq = tid % RQ, r0 = tid / RQ;
        const int gx = x0 - 4 + 4 * rq;
        unsigned v[IT];
        if (x0 - 4 >= 0 && x0 + kBlurTW + 4 <= gw) {
            // r04: a tile whose window lies inside the plane horizontally (uniform: 8 of 10 tile columns at level 0) needs no
            // per-byte reflection -- one buffer load per dword with a 32-bit offset (the staging of the general path below was
            // ~300 of the kernel's ~510 VALU instructions per thread: clamps, selects and 64-bit addresses for six loads)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, pitch * gh, 0x00020000);
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int ry = min(r0 + RPP * k, RH - 1);
                int gy = y0 - 3 + ry; gy = gy < 0 ? -gy : gy; gy = gy >= gh ? 2 * (gh - 1) - gy : gy; gy = max(gy, 0);
                v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, gy * pitch + gx, 0, 0);
            }
        } else {
            // an edge tile (the first / last tile column of a level: 2 of 10 at level 0, 2 of 3 at level 7).  r06: branch-free -- EVERY lane gathers its four bytes at
            // reflected columns (for an inner lane they are its dword), all rows at once.  The form this replaces (`inner ? dword load : four reflected byte loads` per row)
            // compiled to two waits per row: up to twelve sequential round trips per wave in an edge tile against one in an inner tile
            int xb[4];
#pragma unroll
            for (int b = 0; b < 4; b++) { int x = gx + b; x = x < 0 ? -x : x; x = x >= gw ? 2 * (gw - 1) - x : x; xb[b] = min(max(x, 0), gw - 1); }
            unsigned bt[IT][4];
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int ry = min(r0 + RPP * k, RH - 1);
                int gy = y0 - 3 + ry; gy = gy < 0 ? -gy : gy; gy = gy >= gh ? 2 * (gh - 1) - gy : gy; gy = max(gy, 0);
                const uint8_t* row = src + (size_t)gy * pitch;
#pragma unroll
                for (int b = 0; b < 4; b++) bt[k][b] = row[xb[b]];
            }
#pragma unroll
            for (int k = 0; k < IT; k++) asm volatile("" : "+v"(bt[k][0]), "+v"(bt[k][1]), "+v"(bt[k][2]), "+v"(bt[k][3]));     // (all in flight before the first use)
#pragma unroll
            for (int k = 0; k < IT; k++) v[k] = bt[k][0] | (bt[k][1] << 8) | (bt[k][2] << 16) | (bt[k][3] << 24);
        }
#pragma unroll
        for (int k = 0; k < IT; k++) if (r0 < RPP && r0 + RPP * k < RH) raw[(r0 + RPP * k) * RQ + rq] = v[k];
    } else {
        for (int i = tid; i < RH * RQ; i += 256) {                   // tiny planes: general reflection
            const int ry = i / RQ, rq = i % RQ;
            const int gy = reflect101(y0 - 3 + ry, gh), gx = x0 - 4 + 4 * rq;
            const uint8_t* row = src + (size_t)gy * pitch;
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) w |= (unsigned)row[reflect101(gx + b, gw)] << (8 * b);
            raw[i] = w;
        }
    }
    __syncthreads();
    // coefficients as bytes: taps -3..0 and +1..+3 (A-4 table or its <= 3.4.1 form)
    const unsigned k2 = cfg->varBlur ? 49u : 48u, k3 = cfg->varBlur ? 55u : 56u;
    const unsigned K1 = 18u | (34u << 8) | (k2 << 16) | (k3 << 24), K2 = k2 | (34u << 8) | (18u << 16);
    // horizontal pass: item = (row pair, quad); results of the two rows packed per column for the vertical pass's v_dot2
    for (int i = tid; i < (RH / 2) * HQ; i += 256) {
        const int p = i / HQ, q = i % HQ;
        unsigned h[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const unsigned* rp = raw + (2 * p + rr) * RQ + q;
            const unsigned w0 = rp[0], w1 = rp[1], w2 = rp[2];   // pixels x-4.., x.., x+4.. (x = x0 + 4q)
            h[rr][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), K2, 0u, false), false);
            h[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), K2, 0u, false), false);
            h[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), K2, 0u, false), false);
            h[rr][3] = __builtin_amdgcn_udot4(w1, K1, __builtin_amdgcn_udot4(w2, K2, 0u, false), false);
        }
        hp[i] = make_uint4(h[0][0] | (h[1][0] << 16), h[0][1] | (h[1][1] << 16), h[0][2] | (h[1][2] << 16), h[0][3] | (h[1][3] << 16));
    }
    __syncthreads();
    // vertical pass: thread = quad q (4 columns) x 4 consecutive output rows oy0 .. oy0+3 = hp rows oy0 .. oy0+9 = 5 row pairs.
    // Output row oy0 + r reads hp rows oy0+r .. oy0+r+6: for even r the pairs (k0,k1)(k2,k3)(k4,k5)(k6,0) from pair r/2 on, for
    // odd r (0,k0)(k1,k2)(k3,k4)(k5,k6) from pair (r-1)/2 on -- 4 v_dot2_u32_u16 per pixel instead of 7 multiply-adds.
    uint8_t* dst = blur + (size_t)img * cfg->pyrBytes + G.off;
    {
        const int q = tid & (HQ - 1), rg = tid / HQ;                    // 32 quads x 8 row groups = 256 threads
        const int oy0 = 4 * rg;
        const unsigned KE[4] = {18u | (34u << 16), k2 | (k3 << 16), k2 | (34u << 16), 18u};
        const unsigned KO[4] = {18u << 16, 34u | (k2 << 16), k3 | (k2 << 16), 34u | (18u << 16)};
        uint4 P[5];
#pragma unroll
        for (int j = 0; j < 5; j++) P[j] = hp[(2 * rg + j) * HQ + q];
        if (x0 + 4 * q < G.pitch) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int y = y0 + oy0 + r;
                if (y >= G.h) break;
                unsigned acc[4] = {32768u, 32768u, 32768u, 32768u};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint4 pj = P[(r >> 1) + j];
                    const unsigned kk = (r & 1) ? KO[j] : KE[j];
                    acc[0] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pj.x), __builtin_bit_cast(u16x2, kk), acc[0], false);
                    acc[1] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pj.y), __builtin_bit_cast(u16x2, kk), acc[1], false);
                    acc[2] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pj.z), __builtin_bit_cast(u16x2, kk), acc[2], false);
                    acc[3] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pj.w), __builtin_bit_cast(u16x2, kk), acc[3], false);
                }
                // (acc >> 16) saturated to 255: clamp to 0x00ffffff, then byte 2 of each sum
                const unsigned a0 = min(acc[0], 0xffffffu), a1 = min(acc[1], 0xffffffu), a2 = min(acc[2], 0xffffffu), a3 = min(acc[3], 0xffffffu);
                const unsigned lo = __builtin_amdgcn_perm(a1, a0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(a3, a2, 0x06020c0cu);
                *(unsigned*)(dst + (size_t)y * G.pitch + x0 + 4 * q) = lo | hi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// libstdc++ std::nth_element (bits/stl_algo.h __introselect) on 64-bit keys whose high word is the
// response as f32 bits (responses are >= 0, so unsigned compare == float compare).  Must replay the
// exact compare/swap sequence: FAST responses tie constantly and the surviving ORDER feeds every
// downstream index (SURVEY §7 hard part 1).  Comparator = cv::KeypointResponseGreater.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
#define RGT(a, b) ((unsigned)((a) >> 32) > (unsigned)((b) >> 32))

template <typename PTR>
DEVINL void sel_adjust_heap(PTR f, int hole, int len, u64 value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (RGT(f[child], f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && RGT(f[parent], value)) {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = value;
}
template <typename PTR>
DEVINL void sel_nth_element(PTR v, int n, int nth)
{
    if (n <= 0 || nth >= n) return;
    int first = 0, last = n;
    int depth = 0;
    for (int t = n; t > 1; t >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            // __heap_select(first, nth+1, last) ; iter_swap(first, nth)
            PTR f = v + first;
            const int len = nth + 1 - first;
            if (len >= 2) {
                int parent = (len - 2) / 2;
                for (;;) {
                    u64 val = f[parent];
                    sel_adjust_heap(f, parent, len, val);
                    if (parent == 0) break;
                    parent--;
                }
            }
            for (int i = nth + 1; i < last; ++i)
                if (RGT(v[i], f[0])) { u64 val = v[i]; v[i] = f[0]; sel_adjust_heap(f, 0, len, val); }
            u64 tmp = v[first]; v[first] = v[nth]; v[nth] = tmp;
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        // __move_median_to_first(first, first+1, mid, last-1)
        {
            const int a = first + 1, b = mid, c = last - 1;
            u64 va = v[a], vb = v[b], vc = v[c];
            int m;
            if (RGT(va, vb)) { if (RGT(vb, vc)) m = b; else if (RGT(va, vc)) m = c; else m = a; }
            else if (RGT(va, vc)) m = a;
            else if (RGT(vb, vc)) m = c;
            else m = b;
            u64 tmp = v[first]; v[first] = v[m]; v[m] = tmp;
        }
        // __unguarded_partition(first+1, last, pivot = first)
        int lo = first + 1, hi = last;
        const u64 pivot = v[first];
        for (;;) {
            while (RGT(v[lo], pivot)) ++lo;
            --hi;
            while (RGT(pivot, v[hi])) --hi;
            if (!(lo < hi)) break;
            u64 tmp = v[lo]; v[lo] = v[hi]; v[hi] = tmp;
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    // __insertion_sort(first, last)
    for (int i = first + 1; i < last; ++i) {
        const u64 val = v[i];
        if (RGT(val, v[first])) {
            for (int k = i; k > first; --k) v[k] = v[k - 1];
            v[first] = val;
        } else {
            int l = i;
            while (RGT(val, v[l - 1])) { v[l] = v[l - 1]; --l; }
            v[l] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative replay of the same introselect.  The sequential bottleneck of std::nth_element is
// __unguarded_partition's two pointer scans; its outcome, however, is a pure function of the array at
// the start of the round:
//   a_0 < a_1 < ...  positions >= first+1 whose value is NOT greater than the pivot  (where `lo` stops)
//   b_0 > b_1 > ...  positions <= last-1  whose value is NOT less    than the pivot  (where `hi` stops)
//   the loop swaps (a_k, b_k) for k < K, K = first k with a_k >= b_k, and returns
//   cut = min(a_K, b_{K-1})   (a swapped-in value <= pivot sits at b_{K-1}; b_{-1} = +inf)
// so one round is two ordered compactions (wave ballots), a count and K independent swaps: O(range/64)
// steps instead of O(range).  median-of-3, the <= 3 element insertion sort and the depth-limit heap-select
// fallback stay on lane 0.  All 64 lanes call this with uniform arguments; v, A, B live in LDS.
// ------------------------------------------------------------------------------------------------
// GLOBAL = the lists live in global memory (k_cell_select_huge): order the wave's own stores and loads with a
// workgroup-scope fence (s_waitcnt vmcnt(0)); the CU's L1 is write-through, so the wave then reads what it wrote
template <bool GLOBAL> DEVINL void wave_sync_mem()
{
    if constexpr (GLOBAL) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else wave_sync_lds();
}
template <typename PTR, typename IDX, bool GLOBAL = false>
DEVINL void sel_nth_element_wave(PTR v, int n, int nth, IDX A, IDX B, int lane)
{
#define wave_sync_lds wave_sync_mem<GLOBAL>
    if (n <= 0 || nth >= n) return;
    int first = 0, last = n;
    int depth = 0;
    for (int t = n; t > 1; t >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {                       // rare: finish sequentially exactly as libstdc++ does
            if (lane == 0) {
                PTR f = v + first;
                const int len = nth + 1 - first;
                if (len >= 2) {
                    int parent = (len - 2) / 2;
                    for (;;) { u64 val = f[parent]; sel_adjust_heap(f, parent, len, val); if (parent == 0) break; parent--; }
                }
                for (int i = nth + 1; i < last; ++i)
                    if (RGT(v[i], f[0])) { u64 val = v[i]; v[i] = f[0]; sel_adjust_heap(f, 0, len, val); }
                u64 tmp = v[first]; v[first] = v[nth]; v[nth] = tmp;
            }
            wave_sync_lds();
            return;
        }
        --depth;
        if (lane == 0) {                        // __move_median_to_first(first, first+1, mid, last-1)
            const int mid = first + (last - first) / 2;
            const int a = first + 1, b = mid, c = last - 1;
            const u64 va = v[a], vb = v[b], vc = v[c];
            int m;
            if (RGT(va, vb)) { if (RGT(vb, vc)) m = b; else if (RGT(va, vc)) m = c; else m = a; }
            else if (RGT(va, vc)) m = a;
            else if (RGT(vb, vc)) m = c;
            else m = b;
            const u64 tmp = v[first]; v[first] = v[m]; v[m] = tmp;
        }
        wave_sync_lds();
        const unsigned pivot = (unsigned)(v[first] >> 32);
        const int lo0 = first + 1, len = last - lo0;
        // ordered compaction of the left stops (ascending) and right stops (descending position)
        int nA = 0, nB = 0;
        for (int base = 0; base < len; base += 64) {
            const int i = lo0 + base + lane;                     // ascending walk
            const bool inA = base + lane < len && !((unsigned)(v[i] >> 32) > pivot);
            const unsigned long long mA = __ballot(inA);
            if (inA) A[nA + __popcll(mA & ((1ull << lane) - 1ull))] = i;
            nA += __popcll(mA);
            const int j = last - 1 - base - lane;                // descending walk
            const bool inB = base + lane < len && !(pivot > (unsigned)(v[j] >> 32));
            const unsigned long long mB = __ballot(inB);
            if (inB) B[nB + __popcll(mB & ((1ull << lane) - 1ull))] = j;
            nB += __popcll(mB);
        }
        wave_sync_lds();
        // K = number of leading pairs with a_k < b_k (monotone predicate)
        const int nP = min(nA, nB);
        int K = 0;
        for (int base = 0; base < nP; base += 64) {
            const int k = base + lane;
            const bool ok = k < nP && (int)A[k] < (int)B[k];
            const unsigned long long mk = __ballot(ok);
            K += __popcll(mk);
            if (mk != ~0ull) break;
        }
        const int aK = K < nA ? (int)A[K] : 0x7fffffff;
        const int bKm1 = K > 0 ? (int)B[K - 1] : 0x7fffffff;
        const int cut = min(aK, bKm1);
        for (int k = lane; k < K; k += 64) {                     // the swaps touch disjoint positions
            const int ia = A[k], ib = B[k];
            const u64 ta = v[ia], tb = v[ib];
            v[ia] = tb; v[ib] = ta;
        }
        wave_sync_lds();
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0) {                            // __insertion_sort(first, last), <= 3 elements
        for (int i = first + 1; i < last; ++i) {
            const u64 val = v[i];
            if (RGT(val, v[first])) { for (int k = i; k > first; --k) v[k] = v[k - 1]; v[first] = val; }
            else { int l = i; while (RGT(val, v[l - 1])) { v[l] = v[l - 1]; --l; } v[l] = val; }
        }
    }
    wave_sync_lds();
#undef wave_sync_lds
}

// ------------------------------------------------------------------------------------------------
// Keypoint selection = ComputeKeyPointsOld :880-1213 minus the per-pixel work, in three kernels:
//   k_quota        per (image, level): `<=3 -> minThFAST` fallback per cell (:1047) from k_fast_nms's counters,
//                  cost-map cell weights and quotas (:946-987, :1028-1031), the single-pass quota redistribution
//                  (:1103-1133); what retainBest will keep per cell is data-independent, so the offsets of the
//                  concatenated level list are fixed here too.
//   k_cell_select  per (image, cell), one wave: restore cv::FAST's row-major order (bitonic sort of the packed
//                  positions in LDS), response x quality (:1058-1080), retainBest (:1146-1148) = exact replay of
//                  libstdc++ introselect by one lane on the LDS list, kept entries -> level list (i,j order).
//   k_level_select per (image, level): level-wide retainBest (:1162-1166), keypoint slots, level count.
// Candidate counts per cell are heavy-tailed (tens typically, >1000 on repetitive texture); one wave per cell lets
// the hardware balance that, and keeps every sequential replay in LDS (~64-cycle steps instead of L2 round trips).
// ------------------------------------------------------------------------------------------------
// level of a global cell index: the last VALID level whose cellBase is <= gc, from one wide scalar load (r06: the loop over cfg->lv[l].valid / .cellBase was two
// dependent scalar round trips per level at the head of every cell's wave)
DEVINL int level_of_cell(const Config* __restrict__ cfg, int gc)
{
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) level = gc >= cfg->cellBases[l] ? l : level;
    return level;
}

struct CellInfo { int nTotal, nRetain, prefix, useMin; };

// cv::sum over every cell WINDOW of the cost pyramid (:977), one wave per cell; the sums are parked in cellInfo[].x
// until k_quota consumes them (introspection only).
__global__ __launch_bounds__(256) void k_cell_qsum(const Config* __restrict__ cfg, const uint8_t* __restrict__ qpyr,
                                                  const uint8_t* __restrict__ useCost, CellInfo* __restrict__ cellInfo)
{
    const int img = blockIdx.y, lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // a scalar: the level geometry below then comes through scalar loads
    if (cell >= cfg->nCellsTotal || !(cfg->introspection && (use_cost_of(useCost, img) & 1))) return;
    const int level = level_of_cell(cfg, cell);
    const LevelGeom& G = cfg->lv[level];
    const int c = cell - G.cellBase, cols = G.cols, rows = G.rows;
    if (!G.valid || c >= G.nCells) return;
    const uint8_t* Q = qpyr + (size_t)img * cfg->pyrBytes + G.off;
    const int i = c / cols, j = c % cols;
    const int x0 = kEdge + j * G.cellW, y0 = kEdge + i * G.cellH;
    const int wx0 = x0 - 3, wx1 = (j == cols - 1) ? G.maxBX + 3 : x0 + G.cellW + 3;
    const int wy0 = y0 - 3, wy1 = wy0 + ((i == rows - 1) ? G.winHLast : G.cellH + 6);
    // aligned dwords covering [wx0, wx1); bytes outside the window are masked; v_sad_u8 sums 4 bytes per op
    const int a0 = wx0 & ~3, nq = (wx1 - a0 + 3) / 4, nrow = wy1 - wy0;
    unsigned acc = 0;
    // r06: eight dwords per lane in flight.  One load per iteration with its use right behind it is one round trip per iteration: a level-0 cell's window is
    // ~4,500 dwords = 70 SEQUENTIAL round trips per wave, which was the kernel's whole time (87 us per 128 cost images, 325 at 1920 x 1200)
    constexpr int U = 8;
    const int total = nq * nrow;
    const int dr = 64 / nq, dq = 64 % nq;                                      // (row, dword) of item t + 64 from those of item t: one division per wave, not one per item
    int rr = lane / nq, qq = lane % nq;
    for (int t0 = lane; t0 < total; t0 += 64 * U) {
        unsigned v[U]; int bxs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool in = t0 + 64 * u < total;
            const int r = in ? rr : nrow - 1, q = in ? qq : nq - 1;             // past the end: the last item again (not counted)
            bxs[u] = a0 + 4 * q;
            v[u] = *(const unsigned*)(Q + (size_t)(wy0 + r) * G.pitch + bxs[u]);
            qq += dq; rr += dr;
            if (qq >= nq) { qq -= nq; rr++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(v[u]));            // (keeps the compiler from sinking a load into the branch of its use)
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned w = v[u];
            const int bx = bxs[u];
            if (bx < wx0) w &= 0xffffffffu << (8 * (wx0 - bx));
            if (bx + 4 > wx1) w &= 0xffffffffu >> (8 * (bx + 4 - wx1));
            if (t0 + 64 * u < total) acc = __builtin_amdgcn_sad_u8(w, 0u, acc);
        }
    }
    const unsigned qs = (unsigned)wave_sum_i32((int)acc);
    if (lane == 0) cellInfo[(size_t)img * cfg->nCellsTotal + cell].nTotal = (int)qs;
}

constexpr int kCellCapTiny = 256, kCellCapSmall = 1024, kCellCapBig = 4096;
__global__ __launch_bounds__(256) void k_quota(const Config* __restrict__ cfg, const int* __restrict__ cellCnt,
                                              const uint8_t* __restrict__ qpyr, const uint8_t* __restrict__ useCost,
                                              CellInfo* __restrict__ cellInfo, int* __restrict__ lvlTotal,
                                              int* __restrict__ hugeCount, int* __restrict__ hugeList, int* __restrict__ tierList,
                                              int tierCap, int* __restrict__ status, int cap)
{
    // r06: the bookkeeping arrays are sized by the launch (cap = the largest level's cell count, rounded up) instead of kMaxCells = 2048 each: 51 KB per workgroup
    // meant three workgroups per CU for 2,048 (level, image) workgroups -- three rounds of a kernel whose time is one thread's sequential pass
    extern __shared__ __attribute__((aligned(16))) int s_quota[];
    int* const s_nIni = s_quota; int* const s_nMin = s_nIni + cap; int* const s_nTotal = s_nMin + cap; int* const s_nRetain = s_nTotal + cap;
    int* const s_prefix = s_nRetain + cap;                    // [cap + 1]
    unsigned* const s_qsum = (unsigned*)(s_prefix + cap + 4);
    float* const s_diff = (float*)(s_qsum + cap);             // nfeatures_cell - nKeys of the cells that keep all their keypoints
    unsigned char* const s_useMin = (unsigned char*)(s_diff + cap);
    __shared__ float s_wsum;
    const int img = blockIdx.y, level = blockIdx.x;
    const LevelGeom& G = cfg->lv[level];
    const int tid = threadIdx.x;
    if (!G.valid) { if (tid == 0) lvlTotal[img * kMaxLevels + level] = 0; return; }
    const int mode = (cfg->introspection && (use_cost_of(useCost, img) & 1)) ? 1 : 0;
    const int nCells = G.nCells, cols = G.cols, rows = G.rows;
    const int* cnt = cellCnt + ((size_t)img * cfg->nCellsTotal + G.cellBase) * 2;
    CellInfo* ci = cellInfo + (size_t)img * cfg->nCellsTotal + G.cellBase;
    if (mode && G.winHLast <= 0) {
        // the last cell row starts exactly at the bottom border: its hY is 0, the row is skipped (:953-954) and the stale hY
        // makes every other window of the level empty too -- the level yields nothing (accepted geometry, Context::build)
        for (int c = tid; c < nCells; c += 256) ci[c] = CellInfo{0, 0, 0, 0};
        if (tid == 0) lvlTotal[img * kMaxLevels + level] = 0;
        return;
    }
    for (int c = tid; c < nCells; c += 256) {
        s_nMin[c] = cnt[2 * c]; s_nIni[c] = cnt[2 * c + 1];
        s_qsum[c] = mode ? (unsigned)ci[c].nTotal : 0u;       // window sums from k_cell_qsum
    }
    __syncthreads();
    // r06: everything that is a function of ONE cell runs on all threads (the window weight with its f64 division, nfeatures_cell with its division and ceil,
    // the keep-all / retain decision); thread 0 keeps only what the reference's order defines: the float sum of the weights in (i, j) order, the int / float
    // accumulation of nToDistribute, the redistribution pass and the prefix.  (Was: all of it on thread 0 -- ~600 cycles per cell with 255 threads idle.)
    if (mode) {
        for (int c = tid; c < nCells; c += 256) {
            const int i = c / cols, j = c % cols;
            const float hX = (j == cols - 1) ? (float)(G.maxBX + 3 - (16 + j * G.cellW)) : (float)(G.cellW + 6);
            const float hY = (i == rows - 1) ? (float)G.winHLast : (float)(G.cellH + 6);
            const float cost = (float)s_qsum[c] / (float)(hX * hY);
            const float q = (float)(1.0 / (1.0 + (double)(cost / 255)));
            s_qsum[c] = __float_as_uint(2 * q - 1);
        }
        __syncthreads();
        if (tid == 0) { float wsum = 0.0f; for (int c = 0; c < nCells; c++) wsum += __uint_as_float(s_qsum[c]); s_wsum = wsum; }
        __syncthreads();
    }
    {
        const float wsum = mode ? s_wsum : 1.0f;
        for (int c = tid; c < nCells; c += 256) {
            const float nfc = mode ? fmaxf(1.0f, ceilf((float)G.nDesired * __uint_as_float(s_qsum[c]) / wsum))
                                   : (float)G.nfeaturesCell;
            const bool useMin = s_nIni[c] <= 3;
            const int nKeys = useMin ? s_nMin[c] : s_nIni[c];
            s_useMin[c] = useMin;
            s_nTotal[c] = nKeys;
            if ((float)nKeys > nfc) { s_nRetain[c] = (int)nfc; s_prefix[c] = 0; }
            else { s_nRetain[c] = nKeys; s_diff[c] = nfc - (float)nKeys; s_prefix[c] = 1; }      // s_prefix doubles as bNoMore here
            s_nIni[c] = __float_as_int(nfc);            // keep nfeatures_cell for the redistribution pass
        }
    }
    __syncthreads();
    if (tid == 0) {                                   // the order-dependent rest, exactly in (i,j) order
        int nToDistribute = 0, nNoMore = 0;
        for (int c = 0; c < nCells; c++)
            if (s_prefix[c]) { nToDistribute = (int)((float)nToDistribute + s_diff[c]); nNoMore++; }
        if (nToDistribute > 0 && nNoMore < nCells) {
            for (int c = 0; c < nCells; c++) {
                if (!s_prefix[c]) {
                    const int nNew = (int)(__int_as_float(s_nIni[c]) + ceilf((float)nToDistribute / (nCells - nNoMore)));
                    if (s_nTotal[c] > nNew) s_nRetain[c] = nNew;
                    else { s_nRetain[c] = s_nTotal[c]; nToDistribute += nNew - s_nTotal[c]; s_prefix[c] = 1; nNoMore++; }
                }
            }
        }
        int acc = 0;
        for (int c = 0; c < nCells; c++) {
            const int nT = s_nTotal[c], nR = s_nRetain[c];
            s_prefix[c] = acc;
            acc += (nR >= 0 && nT > nR) ? nR : nT;
        }
        lvlTotal[img * kMaxLevels + level] = acc;
    }
    __syncthreads();
    for (int c = tid; c < nCells; c += 256) {
        ci[c] = CellInfo{s_nTotal[c], s_nRetain[c], s_prefix[c], (int)s_useMin[c]};
        // cells above the one-wave tier go on the work list of their tier (hugeCount[1], [2]: r04 -- their kernels walk the list
        // instead of launching a workgroup per cell of every image that looks at its count and leaves)
        if (s_nTotal[c] > kCellCapTiny && s_nTotal[c] <= kCellCapBig) {
            const int tier = s_nTotal[c] > kCellCapSmall ? 1 : 0;
            const int k = atomicAdd(hugeCount + 1 + tier, 1);
            if (k < tierCap) tierList[(size_t)tier * tierCap + k] = img * cfg->nCellsTotal + G.cellBase + c;
            else atomicOr(status, 4);
        }
        if (s_nTotal[c] > kCellCapBig) {                 // too many survivors for the LDS selection: k_cell_select_huge's work list
            const int k = atomicAdd(hugeCount, 1);
            if (hugeList && k < kHugeListCap) hugeList[k] = img * cfg->nCellsTotal + G.cellBase + c;
            else atomicOr(status, 4);
        }
    }
}

// pass 0 handles cells with <= kCellCapSmall candidates (12 KB LDS, many waves per CU); pass 1 those up to kCellCapBig;
// anything bigger (large images with few features: 1920x1200 at N = 500 has 627x290-pixel cells) goes to k_cell_select_huge
// NTHR = 64: one wave per cell (the common, small cells: many cells per CU).  NTHR = 256 for the big-cell pass: gather and
// bitonic sort run on four waves (with the introspection quirk of overlapping cell domains most level-0/1 cells hold
// 1-4 thousand candidates and a single wave spent ~100 us per cell in the sort); the introselect stays on wave 0.
constexpr int kSelRows = 512;                            // tallest cell (rows) the counting sort of k_cell_select handles
template <int CAP>
struct __attribute__((aligned(16))) SelShared {
    unsigned keys[CAP];
    u64 ord[CAP];
    int row[kSelRows + 2];                                    // per-row survivor counts, then their exclusive prefix
    int m;
};
template <int CAP, int NTHR>
DEVINL void cell_select_one(const Config* __restrict__ cfg, const unsigned* __restrict__ tileList,
                            const int* __restrict__ tileCnt, const CellInfo* __restrict__ cellInfo,
                            const uint8_t* __restrict__ qpyr, const uint8_t* __restrict__ useCost,
                            u64* __restrict__ lvlList, int* __restrict__ status, int nLo, int nHi, int img, int gc, int tid, SelShared<CAP>& S)
{
    unsigned* const keys = S.keys;
    u64* const ord = S.ord;
    int* const s_row = S.row;
    int& s_m = S.m;
    unsigned short* stopA = (unsigned short*)keys;            // the sort keys are dead once `ord` is built:
    unsigned short* stopB = stopA + CAP;                      // their space holds the partition stop lists
    const int lane = tid & 63;
    auto sync = [&]() {
        if constexpr (NTHR == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    const int level = level_of_cell(cfg, gc);
    const LevelGeom& G = cfg->lv[level];
    const int c = gc - G.cellBase;
    if (!G.valid || c < 0 || c >= G.nCells) return;
    const CellInfo info = cellInfo[(size_t)img * cfg->nCellsTotal + gc];
    const int nT = info.nTotal, nR = info.nRetain;
    if (nT <= 0) return;
    if (nT <= nLo || nT > nHi) return;                               // this launch's tier: cells with nLo < survivors <= nHi
    const int mode = (cfg->introspection && (use_cost_of(useCost, img) & 1)) ? 1 : 0;
    const uint8_t* Q = mode ? qpyr + (size_t)img * cfg->pyrBytes + G.off : nullptr;
    const unsigned th = info.useMin ? 1u : (unsigned)cfg->iniTh;
    const int kept = (nR >= 0 && nT > nR) ? nR : nT;
    u64* dst = lvlList + (size_t)img * cfg->candTotal + G.candBase + info.prefix;
    if (nT > CAP) return;                                            // > kCellCapBig survivors: k_cell_select_huge
#ifdef IVF_SEL_TIMING
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    // a) collect the cell's survivors from the FAST tiles it overlaps, filtered by its rectangle and threshold
    //    (order irrelevant here)
    const int ci = c / G.cols, cj = c % G.cols;
    const int cx0 = kEdge + cj * G.cellW, cx1 = (cj == G.cols - 1) ? G.maxBX : cx0 + G.cellW;
    const int cy0 = kEdge + ci * G.cellH, cy1 = cy0 + ((ci == G.rows - 1) ? G.domHLast : G.domH[mode]);
    int m = 0;
    if (tid == 0) s_m = 0;
    sync();
    if (cy1 > cy0 && cx1 > cx0) {
        // The gather is a chain of memory round trips (a tile's count, then its list, 64 entries per step): the counts of up to
        // 64 tiles are fetched by one wave-wide load and handed out by lane broadcast, and a tile's list is read four steps at
        // a time (four loads in flight per lane) -- one round trip per tile instead of ~5.  (Measured: 268 -> 263 us per 256 images only -- the kernel's time is the bitonic sort,
        // 41 %, and the introselect replay, 26 %, found by compiling each out.)
        const int tx0 = (cx0 - 16) / kFastTW, tx1 = (cx1 - 1 - 16) / kFastTW;
        const int ty0 = (cy0 - kEdge) / kFastTH, ty1 = (cy1 - 1 - kEdge) / kFastTH;
        const int ntx = tx1 - tx0 + 1, nt = ntx * (ty1 - ty0 + 1);
        const size_t tileBase = (size_t)img * cfg->nTiles + G.tileBase;
        // r04: the lists of up to 64 tiles are read as ONE flattened sequence (a cell of a plain extraction spans 20-30 tiles with a few
        // dozen survivors each: one round trip per tile was 30-40 us of dependent latency per cell).  Lane j holds tile j's count; the
        // exclusive prefix and the tiles' list offsets go to LDS (s_row is free until the sort), every thread finds the tile of its
        // flattened index by a 6-step binary search and U loads per thread are in flight together.
        constexpr int U = 4;
        int* const s_pref = s_row;                // [65]
        int* const s_toff = s_row + 80;           // [64] tile slot index
        for (int t0 = 0; t0 < nt; t0 += 64) {
            int myCnt = 0, myTile = 0;
            if (t0 + lane < nt) {
                const int t = t0 + lane;
                myTile = (ty0 + t / ntx) * G.tilesX + tx0 + t % ntx;
                myCnt = min(tileCnt[tileBase + myTile], kTileCap);
            }
            int incl = myCnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            const int total = __shfl(incl, 63, 64);
            sync();                               // the previous round's readers are done with s_pref / s_toff
            if (tid < 64) { s_pref[lane] = incl - myCnt; s_toff[lane] = myTile; if (lane == 63) s_pref[64] = total; }
            sync();
            for (int f0 = 0; f0 < total; f0 += NTHR * U) {
                unsigned ev[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int f = f0 + u * NTHR + tid;
                    ev[u] = 0u;                   // score 0 < th: never kept
                    if (f < total) {
                        int j = 0;
#pragma unroll
                        for (int st = 32; st > 0; st >>= 1) if (s_pref[j + st] <= f) j += st;      // s_pref[64] = total > f: j + st <= 63 always holds where it matters
                        ev[u] = tileList[(tileBase + s_toff[j]) * kTileCap + (f - s_pref[j])];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (f0 + u * NTHR >= total) break;               // uniform
                    const unsigned e = ev[u];
                    const int ex = (e >> 8) & 0xfff, ey = e >> 20;
                    const bool keep = (e & 0xffu) >= th && ex >= cx0 && ex < cx1 && ey >= cy0 && ey < cy1;
                    const unsigned long long mask = __ballot(keep);
                    if constexpr (NTHR == 64) {
                        if (keep) {
                            const int idx = m + __popcll(mask & ((1ull << lane) - 1ull));
                            if (idx < CAP) keys[idx] = e;
                        }
                        m += __popcll(mask);
                    } else {
                        // order is irrelevant here (the sort restores it): one LDS atomic per wave reserves its slots
                        int base = 0;
                        if (lane == 0 && mask) base = atomicAdd(&s_m, __popcll(mask));
                        base = __shfl(base, 0);
                        if (keep) {
                            const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                            if (idx < CAP) keys[idx] = e;
                        }
                    }
                }
            }
        }
    }
    if constexpr (NTHR != 64) { sync(); m = s_m; }
#ifdef IVF_SEL_TIMING
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0 && m != nT) atomicOr(status, 1);                    // internal consistency
    if (m != nT) return;
    {
        // b) row-major order of the packed positions (y<<20 | x<<8 | score).  A cell is a few dozen rows with a handful of survivors
        //    each, so a two-level counting sort -- rows by histogram + prefix sum, then every survivor's rank among the survivors
        //    of its own row -- costs O(m * row length) comparisons instead of the bitonic network's m log^2 m (which was 41 % of this
        //    kernel); cells taller than kSelRows rows keep the bitonic sort.
        const int nrows = cy1 - cy0;
        if (nrows <= kSelRows) {
            unsigned* arrival = (unsigned*)ord;                      // `ord` is free until step c): arrival index, grouped copy
            unsigned* grouped = arrival + CAP;
            for (int r = tid; r <= nrows; r += NTHR) s_row[r] = 0;
            sync();
            for (int k = tid; k < m; k += NTHR) arrival[k] = (unsigned)atomicAdd(&s_row[(int)(keys[k] >> 20) - cy0], 1);
            sync();
            if (tid < 64) {                                          // exclusive prefix sum over the rows, 64 at a time
                int carry = 0;
                for (int b0 = 0; b0 <= nrows; b0 += 64) {
                    const int r = b0 + lane;
                    const int c = r < nrows ? s_row[r] : 0;
                    int incl = c;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                    if (r <= nrows) s_row[r] = carry + incl - c;
                    carry += __shfl(incl, 63, 64);
                }
            }
            sync();
            for (int k = tid; k < m; k += NTHR) { const unsigned e = keys[k]; grouped[s_row[(int)(e >> 20) - cy0] + arrival[k]] = e; }
            sync();
            for (int p = tid; p < m; p += NTHR) {
                const unsigned e = grouped[p];
                const int r = (int)(e >> 20) - cy0;
                const int b = s_row[r], en = s_row[r + 1];
                int rank = 0;
                for (int j = b; j < en; j++) rank += grouped[j] < e ? 1 : 0;      // same row: the order of x (positions are unique)
                keys[b + rank] = e;
            }
            sync();
        } else {
            int n2 = 64;
            while (n2 < m) n2 <<= 1;
            for (int k = m + tid; k < n2; k += NTHR) keys[k] = 0xffffffffu;
            sync();
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2 / 2; i += NTHR) {
                        const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                        const int r = l | j;
                        const unsigned a = keys[l], b = keys[r];
                        const bool up = (l & k) == 0;
                        if ((a > b) == up) { keys[l] = b; keys[r] = a; }
                    }
                    sync();
                }
        }
#ifdef IVF_SEL_TIMING
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
        // c) 64-bit keys: response (x quality factor) | y | x
        for (int k = tid; k < m; k += NTHR) {
            const unsigned e = keys[k];
            const unsigned y = e >> 20, x = (e >> 8) & 0xfffu;
            float resp = (float)(e & 0xffu);
            if (mode) {
                const float cost = (float)Q[(size_t)y * G.pitch + x];
                resp *= 2 * (1.0f / (1.0f + cost / 255.0f)) - 1;
            }
            ord[k] = ((u64)__float_as_uint(resp) << 32) | (y << 16) | x;
        }
        sync();
#ifdef IVF_SEL_TIMING
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
        // d) retainBest (wave 0; the stop lists overlay `keys`, which every wave has finished reading)
        if (nR > 0 && nT > nR && tid < 64) sel_nth_element_wave(ord, nT, nR - 1 + cfg->varRetain, stopA, stopB, lane);
        sync();
#ifdef IVF_SEL_TIMING
        const unsigned long long st4 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && (img == 3 || img == 100) && (gc % 7) == 0)
            printf("cell %d img %d CAP %d: nT %d nR %d rows %d | gather %llu sort %llu keys %llu select %llu cycles\n", gc, img, CAP, nT, nR, cy1 - cy0, st1 - st0, st2 - st1, st3 - st2, st4 - st3);
#endif
        for (int k = tid; k < kept; k += NTHR) dst[k] = ord[k];
    }
}
// tier 0 (<= kCellCapTiny survivors: in plain ORB extraction every cell): one wave per cell, WPB cells per workgroup -- a quarter of
// the workgroups to dispatch for the same waves in flight
template <int CAP, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_cell_select(const Config* __restrict__ cfg, const unsigned* __restrict__ tileList,
                                                        const int* __restrict__ tileCnt, const CellInfo* __restrict__ cellInfo,
                                                        const uint8_t* __restrict__ qpyr, const uint8_t* __restrict__ useCost,
                                                        u64* __restrict__ lvlList, int* __restrict__ status, int nLo, int nHi)
{
    __shared__ SelShared<CAP> S[WPB];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gc = blockIdx.x * WPB + wave;
    if (gc >= cfg->nCellsTotal) return;
    cell_select_one<CAP, 64>(cfg, tileList, tileCnt, cellInfo, qpyr, useCost, lvlList, status, nLo, nHi, blockIdx.y, gc, threadIdx.x & 63, S[wave]);
}
// tiers 1, 2: the workgroups walk k_quota's work list of this tier (img * nCellsTotal + cell)
template <int CAP, int NTHR>
__global__ __launch_bounds__(NTHR) void k_cell_select_list(const Config* __restrict__ cfg, const unsigned* __restrict__ tileList,
                                                        const int* __restrict__ tileCnt, const CellInfo* __restrict__ cellInfo,
                                                        const uint8_t* __restrict__ qpyr, const uint8_t* __restrict__ useCost,
                                                        u64* __restrict__ lvlList, int* __restrict__ status, int nLo, int nHi,
                                                        const int* __restrict__ count, const int* __restrict__ list, int cap)
{
    __shared__ SelShared<CAP> S;
    const int n = min(*count, cap);
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int e = list[w];
        cell_select_one<CAP, NTHR>(cfg, tileList, tileCnt, cellInfo, qpyr, useCost, lvlList, status, nLo, nHi, e / cfg->nCellsTotal, e % cfg->nCellsTotal,
                                   threadIdx.x, S);
        if constexpr (NTHR == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
}

// Cells with more than kCellCapBig survivors: the same four steps with every list in a global scratch slot (one slot per
// workgroup, kHugeSlots workgroups walk k_quota's work list).  Rare by construction -- it exists so that such an image costs
// time instead of an error.  1024 threads gather and sort; wave 0 replays the introselect.
__global__ __launch_bounds__(1024) void k_cell_select_huge(const Config* __restrict__ cfg, const unsigned* __restrict__ tileList,
                                                          const int* __restrict__ tileCnt, const CellInfo* __restrict__ cellInfo,
                                                          const uint8_t* __restrict__ qpyr, const uint8_t* __restrict__ useCost,
                                                          u64* __restrict__ lvlList, int* __restrict__ status,
                                                          const int* __restrict__ hugeCount, const int* __restrict__ hugeList,
                                                          unsigned* __restrict__ scratch, int slotCap)
{
    __shared__ int s_m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nHuge = min(*hugeCount, kHugeListCap);
    // slot layout: keys u32[2 * slotCap] (bitonic padding), ord u64[slotCap], stop lists u32[slotCap] x 2
    unsigned* keys = scratch + (size_t)blockIdx.x * ((size_t)slotCap * 6);
    u64* ord = (u64*)(keys + 2 * (size_t)slotCap);
    unsigned* stopA = (unsigned*)(ord + slotCap);
    unsigned* stopB = stopA + slotCap;
    for (int w = blockIdx.x; w < nHuge; w += gridDim.x) {
        const int img = hugeList[w] / cfg->nCellsTotal, gc = hugeList[w] % cfg->nCellsTotal;
        const int level = level_of_cell(cfg, gc);
        const LevelGeom& G = cfg->lv[level];
        const int c = gc - G.cellBase;
        const CellInfo info = cellInfo[(size_t)img * cfg->nCellsTotal + gc];
        const int nT = info.nTotal, nR = info.nRetain;
        const int mode = (cfg->introspection && (use_cost_of(useCost, img) & 1)) ? 1 : 0;
        const uint8_t* Q = mode ? qpyr + (size_t)img * cfg->pyrBytes + G.off : nullptr;
        const unsigned th = info.useMin ? 1u : (unsigned)cfg->iniTh;
        const int kept = (nR >= 0 && nT > nR) ? nR : nT;
        u64* dst = lvlList + (size_t)img * cfg->candTotal + G.candBase + info.prefix;
        __syncthreads();                                 // previous cell done with the slot and with s_m
        if (nT > slotCap) { if (tid == 0) atomicOr(status, 4); continue; }     // cannot happen: slotCap = max strict maxima per cell
        const int ci = c / G.cols, cj = c % G.cols;
        const int cx0 = kEdge + cj * G.cellW, cx1 = (cj == G.cols - 1) ? G.maxBX : cx0 + G.cellW;
        const int cy0 = kEdge + ci * G.cellH, cy1 = cy0 + ((ci == G.rows - 1) ? G.domHLast : G.domH[mode]);
        if (tid == 0) s_m = 0;
        __syncthreads();
        const int tx0 = (cx0 - 16) / kFastTW, tx1 = (cx1 - 1 - 16) / kFastTW;
        const int ty0 = (cy0 - kEdge) / kFastTH, ty1 = (cy1 - 1 - kEdge) / kFastTH;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const size_t tile = (size_t)img * cfg->nTiles + G.tileBase + ty * G.tilesX + tx;
                const int nAll = min(tileCnt[tile], kTileCap);
                const unsigned* in = tileList + tile * kTileCap;
                for (int b0 = 0; b0 < nAll; b0 += 1024) {
                    const int k = b0 + tid;
                    const unsigned e = k < nAll ? in[k] : 0u;
                    const int ex = (e >> 8) & 0xfff, ey = e >> 20;
                    const bool keep = k < nAll && (e & 0xffu) >= th && ex >= cx0 && ex < cx1 && ey >= cy0 && ey < cy1;
                    const unsigned long long mask = __ballot(keep);
                    int base = 0;
                    if (lane == 0 && mask) base = atomicAdd(&s_m, __popcll(mask));
                    base = __shfl(base, 0);
                    if (keep) {
                        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                        if (idx < slotCap) keys[idx] = e;
                    }
                }
            }
        __syncthreads();
        const int m = s_m;
        if (m != nT) { if (tid == 0) atomicOr(status, 1); continue; }
        int n2 = 64;
        while (n2 < m) n2 <<= 1;
        for (int k = m + tid; k < n2; k += 1024) keys[k] = 0xffffffffu;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2 / 2; i += 1024) {
                    const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const int r = l | j;
                    const unsigned a = keys[l], b = keys[r];
                    const bool up = (l & k) == 0;
                    if ((a > b) == up) { keys[l] = b; keys[r] = a; }
                }
                __syncthreads();
            }
        for (int k = tid; k < m; k += 1024) {
            const unsigned e = keys[k];
            const unsigned y = e >> 20, x = (e >> 8) & 0xfffu;
            float resp = (float)(e & 0xffu);
            if (mode) {
                const float cost = (float)Q[(size_t)y * G.pitch + x];
                resp *= 2 * (1.0f / (1.0f + cost / 255.0f)) - 1;
            }
            ord[k] = ((u64)__float_as_uint(resp) << 32) | (y << 16) | x;
        }
        __syncthreads();
        if (nR > 0 && nT > nR && tid < 64) sel_nth_element_wave<u64*, unsigned*, true>(ord, nT, nR - 1 + cfg->varRetain, stopA, stopB, lane);
        __syncthreads();
        for (int k = tid; k < kept; k += 1024) dst[k] = ord[k];
    }
}

__global__ __launch_bounds__(256) void k_level_select(const Config* __restrict__ cfg, const int* __restrict__ lvlTotal,
                                                     u64* __restrict__ lvlList, unsigned* __restrict__ slotPos,
                                                     float* __restrict__ slotResp, int* __restrict__ lvlCount, int LCAP)
{
    // r06: the LDS copy of a level's list is sized by the launch (LCAP = four times the largest nDesired of the geometry, at most 4096; a longer list takes the
    // global path below as before) instead of 4096 entries = 49 KB per workgroup = three workgroups per CU for 2,048 (level, image) workgroups
    extern __shared__ __attribute__((aligned(16))) u64 s_list[];
    unsigned short* const s_stopA = (unsigned short*)(s_list + LCAP);
    unsigned short* const s_stopB = s_stopA + LCAP;
    const int img = blockIdx.y, level = blockIdx.x, tid = threadIdx.x;
    const LevelGeom& G = cfg->lv[level];
    if (!G.valid) { if (tid == 0) lvlCount[img * kMaxLevels + level] = 0; return; }
    int total = lvlTotal[img * kMaxLevels + level];
    u64* Lg = lvlList + (size_t)img * cfg->candTotal + G.candBase;
    u64* L = Lg;
    if (total > G.nDesired) {
        if (total <= LCAP) {
            for (int k = tid; k < total; k += 256) s_list[k] = Lg[k];
            L = s_list;
            __syncthreads();
        }
        if (L == s_list) { if (tid < 64) sel_nth_element_wave(s_list, total, G.nDesired - 1 + cfg->varRetain, s_stopA, s_stopB, tid); }
        else if (tid == 0) sel_nth_element(L, total, G.nDesired - 1 + cfg->varRetain);   // :1162-1166 (global fallback)
        total = G.nDesired;
        __syncthreads();
    }
    for (int k = tid; k < total; k += 256) {
        const u64 e = L[k];
        slotPos[(size_t)img * cfg->nfeatures + G.kpBase + k] = (unsigned)e;
        slotResp[(size_t)img * cfg->nfeatures + G.kpBase + k] = __uint_as_float((unsigned)(e >> 32));
    }
    if (tid == 0) lvlCount[img * kMaxLevels + level] = total;
}

// ------------------------------------------------------------------------------------------------
// glibc >= 2.28 sinf/cosf (ARM optimized-routines sincosf: f64 polynomial after reduction by pi/2),
// restated so the device rounds exactly like the host libm the reference calls (ORBextractor.cc:113-114).
// ------------------------------------------------------------------------------------------------
DEVINL float sincos_poly(double x, double x2, bool neg, int n)
{
    const double c0 = neg ? -0x1p0 : 0x1p0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = s2 + x2 * s3, x7 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x7 * t1);
    } else {
        const double x4 = x2 * x2, t2 = c3 + x2 * c4, t1 = c0 + x2 * c1, x6 = x4 * x2, c = t1 + x4 * c2;
        return (float)(c + x6 * t2);
    }
}
DEVINL void sincosf_glibc(float y, float& sn, float& cs)
{
    double x = y;
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ff;
    if (top < ((__float_as_uint(0x1.921FB6p-1f) >> 20) & 0x7ff)) {
        if (top < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ff)) { sn = y; cs = 1.0f; return; }
        const double x2 = x * x;
        sn = sincos_poly(x, x2, false, 0);
        cs = sincos_poly(x, x2, false, 1);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const bool neg = (n & 2) != 0;
    const double x2 = x * x;
    sn = sincos_poly(x * sgn, x2, neg, n);
    cs = sincos_poly(x * sgn, x2, neg, n ^ 1);
}

// cv::fastAtan2 (OpenCV 3.x/4.x), f32, degrees
// OpenCV <= 2.4.3: x*y / (x^2 + 0.28 y^2) in double (selected by Config::varAtan)
DEVINL float fast_atan2_legacy(float y, float x)
{
    double a; const double x2 = (double)x * x, y2 = (double)y * y;
    if (y2 <= x2) {
        a = (180. / 3.14159265358979323846) * x * y / (x2 + 0.28 * y2 + 2.2204460492503131e-16);
        return (float)(x < 0 ? a + 180 : y >= 0 ? a : 360 + a);
    }
    a = (180. / 3.14159265358979323846) * x * y / (y2 + 0.28 * x2 + 2.2204460492503131e-16);
    return (float)(y > 0 ? 90 - a : 270 - a);
}
DEVINL float fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
DEVINL int cv_round(float v) { return __float2int_rn(v); }

// ------------------------------------------------------------------------------------------------
// k_describe: one workgroup = kDescKP keypoint slots (4 waves x kDescPW slots).  IC_Angle (:78-105) over the 749-px disc of
// the un-blurred level, rBRIEF (:109-148) on the blurred level, then output assembly (:1253-1294): keypoints of all levels
// concatenated in level order, pt scaled by mvScaleFactor[level] for level>0, plus mvKeyQualScore (Frame.cc:130-143) from
// level 0 of the cost pyramid.
// r01's one-wave-per-keypoint form was bound by memory LATENCY, not by instructions or bytes: a keypoint is two dependent
// round trips (disc -> angle -> rotated samples) over ~68 distinct cache lines, and 32 waves per CU keep only 32 keypoints in
// flight (292 us per 256 images; 184 us with every sample redirected to one cache line).  Here a wave owns kDescPW keypoints
// and issues the loads of ALL of them before it consumes the first: 4 dword loads per keypoint for the disc (a lane reads 4
// adjacent bytes of 4 rows; moments from v_dot4), 8 byte loads per keypoint for the samples.  Between the two phases the
// per-keypoint scalars -- fastAtan2 and glibc's sincosf (~85 f64 instructions) -- are evaluated by ONE lane per keypoint
// instead of redundantly by the 64 lanes of its wave.
// ------------------------------------------------------------------------------------------------
constexpr int kDescPW = 8, kDescKP = 4 * kDescPW;
__global__ __launch_bounds__(256) void k_describe(const Config* __restrict__ cfg, const uint8_t* __restrict__ pyr,
                                                 const uint8_t* __restrict__ blur, const uint8_t* __restrict__ qpyr,
                                                 const uint8_t* __restrict__ useCost,
                                                 const unsigned* __restrict__ slotPos, const float* __restrict__ slotResp,
                                                 const int* __restrict__ lvlCount, ivf_keypoint* __restrict__ kps,
                                                 uint8_t* __restrict__ desc, int* __restrict__ count,
                                                 float* __restrict__ quality, int nImg)
{
    __shared__ int s_m10[kDescKP], s_m01[kDescKP], s_level[kDescKP], s_oi[kDescKP];      // s_level < 0: empty slot
    __shared__ unsigned s_pos[kDescKP];
    __shared__ float s_angle[kDescKP], s_a[kDescKP], s_b[kDescKP];
    // r06: the level's plane offset / pitch / scale / patch size per slot.  `cfg->lv[level]` with `level` read back from LDS is a VECTOR load (the compiler cannot
    // know the wave agrees on it) with an s_waitcnt vmcnt(0) behind it, in front of every keypoint's disc / patch loads: the eight keypoints of a wave, whose loads
    // were meant to be in flight together, went one round trip at a time.  The lane that resolves the slot reads the geometry once.
    __shared__ int s_goff[kDescKP], s_gpitch[kDescKP], s_gpatch[kDescKP];
    __shared__ float s_gscale[kDescKP], s_resp[kDescKP], s_qual[kDescKP];      // + the slot's response and mvKeyQualScore: fetched by the resolving lane too, not
                                                                               // by lane 0 of the wave behind a wait in front of every keypoint's stores
    // image i is described by XCD i % 8 only (its two pyramids, 3.8 MB at 1242 x 375, stay in that XCD's L2)
    const int nf = cfg->nfeatures, nl = cfg->nlevels;
    int img, grp;
    if (!xcd_tile_image((nf + kDescKP - 1) / kDescKP, nImg, grp, img)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int* lc = lvlCount + img * kMaxLevels;
#ifdef IVF_DESC_TIMING
    const unsigned long long dt0 = __builtin_amdgcn_s_memtime();
#endif
    // phase 0: slot -> (level, output index, position), one lane per slot
    if (threadIdx.x < kDescKP) {
        const int slot = grp * kDescKP + threadIdx.x;
        int level = -1, oi = 0;
        unsigned pos = 0;
        if (slot < nf) {
            const unsigned myPos = slotPos[(size_t)img * nf + slot];      // requested together with the level counts (one round trip, not two)
            int lv = 0;
            for (int l = 1; l < nl; l++) if (slot >= cfg->lv[l].kpBase) lv = l;
            int before = 0, total = 0;
            for (int l = 0; l < nl; l++) { const int c = lc[l]; if (l < lv) before += c; total += c; }
            if (slot == 0) count[img] = total;
            const int k = slot - cfg->lv[lv].kpBase;
            if (k < lc[lv]) { level = lv; oi = before + k; pos = myPos; }
            const LevelGeom& Gl = cfg->lv[lv];
            s_goff[threadIdx.x] = Gl.off; s_gpitch[threadIdx.x] = Gl.pitch; s_gpatch[threadIdx.x] = Gl.scaledPatch; s_gscale[threadIdx.x] = Gl.scale;
            s_resp[threadIdx.x] = slotResp[(size_t)img * nf + slot];
            float qv = 1.0f;
            if (level >= 0 && (use_cost_of(useCost, img) & 2)) {          // Frame.cc:130-143: whenever a cost image came with the frame, whatever the extractor flag
                float fx = (float)(myPos & 0xffff), fy = (float)(myPos >> 16);
                if (lv != 0) { fx *= Gl.scale; fy *= Gl.scale; }
                const int qx = (int)roundf(fx), qy = (int)roundf(fy);
                const LevelGeom& G0 = cfg->lv[0];
                const float cost = (float)qpyr[(size_t)img * cfg->pyrBytes + G0.off + (size_t)min(qy, G0.h - 1) * G0.pitch + min(qx, G0.w - 1)];
                const float qs = (float)(1.0 / (1.0 + (double)(cost / 256)));
                qv = 2 * qs - 1;
            }
            s_qual[threadIdx.x] = qv;
        }
        s_level[threadIdx.x] = level; s_oi[threadIdx.x] = oi; s_pos[threadIdx.x] = pos;
    }
    __syncthreads();
#ifdef IVF_DESC_TIMING
    const unsigned long long dt1 = __builtin_amdgcn_s_memtime();
#endif
    const uint8_t* Pimg = pyr + (size_t)img * cfg->pyrBytes;
    const uint8_t* Bimg = blur + (size_t)img * cfg->pyrBytes;
    // phase 1: IC_Angle.  umax (ORBextractor.cc:458-475) depends only on HALF_PATCH_SIZE = 15, so it is a packed constant
    // (4 bits per row; checked against the constructor's table on the host).  Lane = 4 adjacent bytes (u0 .. u0+3,
    // u0 = -15 + 4 (lane & 7)) of the rows v = -15 + (lane >> 3) + 8 q: sum b = dot4(bytes, 1), sum j b = dot4(bytes, (0,1,2,3)),
    // m10 += u0 sum b + sum j b, m01 += v sum b (exact integers: the order of the sum is free).
    {
        const unsigned long long kUmax = 0x3689ABCDDEEEFFFFull;
        const int u0 = -15 + 4 * (lane & 7);
        unsigned w4[kDescPW][4];
#pragma unroll
        for (int r = 0; r < kDescPW; r++) {
            const int ls = wave * kDescPW + r;
            const int level = s_level[ls];
            const unsigned pos = s_pos[ls];
#pragma unroll
            for (int q = 0; q < 4; q++) w4[r][q] = 0;
            if (level >= 0) {                                           // uniform per wave
                const int goff = s_goff[ls], gpitch = s_gpitch[ls];
                const uint8_t* c0 = Pimg + goff + (size_t)(pos >> 16) * gpitch + (pos & 0xffff) + u0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int v = -15 + (lane >> 3) + 8 * q;
                    if (v <= 15) __builtin_memcpy(&w4[r][q], c0 + (ptrdiff_t)v * gpitch, 4);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kDescPW; r++) {
            int m10 = 0, m01 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int v = -15 + (lane >> 3) + 8 * q;
                const int av = v < 0 ? -v : v;
                const int d = v <= 15 ? (int)((kUmax >> (4 * (av & 15))) & 15) : -1;      // |u| <= d inside the disc
                unsigned mask = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { const int u = u0 + j; if (u >= -d && u <= d) mask |= 0xffu << (8 * j); }
                const unsigned bytes = w4[r][q] & mask;
                const int sb = (int)__builtin_amdgcn_udot4(bytes, 0x01010101u, 0u, false);
                const int sj = (int)__builtin_amdgcn_udot4(bytes, 0x03020100u, 0u, false);
                m10 += u0 * sb + sj;
                m01 += v * sb;
            }
            m10 = wave_sum_i32(m10);
            m01 = wave_sum_i32(m01);
            if (lane == 0) { s_m10[wave * kDescPW + r] = m10; s_m01[wave * kDescPW + r] = m01; }
        }
    }
    __syncthreads();
#ifdef IVF_DESC_TIMING
    const unsigned long long dt2 = __builtin_amdgcn_s_memtime();
#endif
    // phase 2: angle, cos, sin -- one lane per keypoint
    if (threadIdx.x < kDescKP && s_level[threadIdx.x] >= 0) {
        const float m01 = (float)s_m01[threadIdx.x], m10 = (float)s_m10[threadIdx.x];
        const float angle = cfg->varAtan ? fast_atan2_legacy(m01, m10) : fast_atan2(m01, m10);
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float a, b;
        sincosf_glibc(angle * factorPI, b, a);
        s_angle[threadIdx.x] = angle; s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    }
    __syncthreads();
#ifdef IVF_DESC_TIMING
    const unsigned long long dt3 = __builtin_amdgcn_s_memtime();
#endif
    // phase 3: rBRIEF (lane handles tests 4*lane .. 4*lane+3).  r04: the 512 samples of a keypoint were 512 single-byte gathers from the
    // blurred plane -- 8 load instructions per lane and keypoint, each touching up to 64 different cache lines (measured with
    // IVF_DESC_TIMING: 56k of a workgroup's 95k cycles, bound by the L1's line requests, not by latency).  Every rotated sample lies within
    // 18.4 px of the keypoint (the pattern's values are in [-13, 13]), so the wave first copies the 39 x 39 patch of each keypoint into
    // LDS with row-contiguous dword loads (39 rows x 11 dwords from the 4-byte boundary left of px - 19: 7 loads per lane instead of 8,
    // and ~9 lines per instruction instead of up to 64) and gathers the samples from there.  Four keypoints of the wave at a time
    // (27 KB of LDS per workgroup).  A keypoint is at least EDGE_THRESHOLD = 19 px inside its level, so the patch is inside the plane.
    const int4 pw = *(const int4*)(d_pattern + lane * 16);
    const int wv[4] = {pw.x, pw.y, pw.z, pw.w};
    constexpr int kPR = 19, kPRows = 2 * kPR + 1, kPDw = 11, kPItems = kPRows * kPDw, kPIt = (kPItems + 63) / 64, kPBytes = kPIt * 64 * 4;
    static_assert(kPR >= kEdge - 0 && kPDw * 4 >= kPRows + 3, "patch row: 39 bytes from a 4-byte boundary at most 3 bytes to the left");
    __shared__ __attribute__((aligned(16))) unsigned s_patch[4][4][kPBytes / 4];
    int prow[kPIt], pcol[kPIt];                 // (row, dword) of this lane's patch items: the same for every keypoint
#pragma unroll
    for (int it = 0; it < kPIt; it++) { const int d = min(lane + 64 * it, kPItems - 1); prow[it] = d / kPDw; pcol[it] = d % kPDw; }
#pragma unroll
    for (int half = 0; half < kDescPW / 4; half++) {
        // (r06, measured and dropped: the patches of all eight keypoints requested at once -- 20 more registers, 220 vs 217-220 us: the second request's round trip
        // is already covered by the other waves' gathers)
        unsigned pv[4][kPIt];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ls = wave * kDescPW + 4 * half + q;
            const int level = s_level[ls];
#pragma unroll
            for (int it = 0; it < kPIt; it++) pv[q][it] = 0u;
            if (level >= 0) {                       // uniform per wave
                const int goff = s_goff[ls], gpitch = s_gpitch[ls];
                const unsigned pos = s_pos[ls];
                const int px = pos & 0xffff, py = pos >> 16;
                const uint8_t* B0 = Bimg + goff + (size_t)(py - kPR) * gpitch + ((px - kPR) & ~3);
#pragma unroll
                for (int it = 0; it < kPIt; it++) pv[q][it] = *(const unsigned*)(B0 + (size_t)prow[it] * gpitch + 4 * pcol[it]);
            }
        }
        wave_sync_lds();                            // the previous half's gathers are done with the patches
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int it = 0; it < kPIt; it++) s_patch[wave][q][lane + 64 * it] = pv[q][it];
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ls = wave * kDescPW + 4 * half + q;
            const int level = s_level[ls];
            if (level < 0) continue;
            const unsigned pos = s_pos[ls];
            const int px = pos & 0xffff, py = pos >> 16;
            const float a = s_a[ls], b = s_b[ls];
            const uint8_t* P = (const uint8_t*)s_patch[wave][q] + kPR * (kPDw * 4) + kPR + ((px - kPR) & 3);      // the keypoint's own pixel
            unsigned nib = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float x0 = (float)(int8_t)(wv[t] & 0xff), y0 = (float)(int8_t)((wv[t] >> 8) & 0xff);
                const float x1 = (float)(int8_t)((wv[t] >> 16) & 0xff), y1 = (float)(int8_t)((wv[t] >> 24) & 0xff);
                const unsigned t0 = P[cv_round(x0 * b + y0 * a) * (kPDw * 4) + cv_round(x0 * a - y0 * b)];
                const unsigned t1 = P[cv_round(x1 * b + y1 * a) * (kPDw * 4) + cv_round(x1 * a - y1 * b)];
                nib |= (unsigned)(t0 < t1) << t;
            }
            const int oi = s_oi[ls];
            unsigned byte = nib | (__shfl_down(nib, 1, 64) << 4);          // valid on even lanes
            unsigned w = byte | (__shfl_down(byte, 2, 64) << 8) | (__shfl_down(byte, 4, 64) << 16) | (__shfl_down(byte, 6, 64) << 24);
            if ((lane & 7) == 0) *(unsigned*)(desc + ((size_t)img * nf + oi) * 32 + (lane >> 3) * 4) = w;
            if (lane == 0) {
                ivf_keypoint kp;
                float fx = (float)px, fy = (float)py;
                const float gscale = s_gscale[ls];
                if (level != 0) { fx *= gscale; fy *= gscale; }
                kp.x = fx; kp.y = fy; kp.size = (float)s_gpatch[ls]; kp.angle = s_angle[ls];
                kp.response = s_resp[ls]; kp.octave = level;
                kps[(size_t)img * nf + oi] = kp;
                const float qv = s_qual[ls];
                quality[(size_t)img * nf + oi] = qv;
            }
        }
    }
#ifdef IVF_DESC_TIMING
    if (threadIdx.x == 0 && (img == 3 || img == 100) && grp == 5) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long dt4 = __builtin_amdgcn_s_memtime();
        printf("describe img %d: lookup %llu disc+moments %llu angle %llu samples+bits+stores %llu cycles\n", img, dt1 - dt0, dt2 - dt1, dt3 - dt2, dt4 - dt3);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// DescriptorDistance (ORBmatcher.cc:1700-1716): 256-bit Hamming = 8 x v_bcnt_u32_b32
// ------------------------------------------------------------------------------------------------
DEVINL int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}
__global__ void k_hamming_pairs(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                const int* __restrict__ pairs, int n, int* __restrict__ dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* pa = (const uint4*)(a + (size_t)pairs[2 * i] * 32);
    const uint4* pb = (const uint4*)(b + (size_t)pairs[2 * i + 1] * 32);
    dist[i] = hamming256(pa[0], pa[1], pb[0], pb[1]);
}

// The pyramid scales of a wave in LDS (r06).  `cfg->scale[octave]` with an octave that came out of a keypoint record is a vector load BEHIND that record -- one
// more dependent round trip per candidate in kernels that are chains of them.  Lane 0 copies the table from scalar loads into the wave's own LDS row (LDS
// operations of one wave execute in order: no workgroup barrier); the lookup is then a conflict-free `ds_read`.  (A local array + a select chain was tried
// first: the compiler promotes it to a per-THREAD LDS copy with 64-byte stride -- 16 KB per workgroup and 16-way bank conflicts on every access.)
DEVINL void stage_scales(const Config* __restrict__ cfg, float* __restrict__ tab, int lane)
{
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < kMaxLevels; l++) tab[l] = cfg->scale[l];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// k_stereo_match: one wave = one left keypoint (Frame.cc:788-915).
//   candidates = right keypoints whose row band [floor(y-r), ceil(y+r)], r = 2*scale[octave], holds
//   row (int)vL, octave within +-1, uR in [uL-maxD, uL]; argmin Hamming (first minimum in increasing
//   iR, as vRowIndices is filled in iR order) starting from TH_HIGH; accept < 75; then the 11x11 SAD
//   over 11 shifts at the keypoint's octave, parabola refinement, disparity gate.
// ------------------------------------------------------------------------------------------------
// k_stereo_rows: the reference's vRowIndices (Frame.cc:775-785) -- for every image row the right keypoints whose band
// [floor(y - r), ceil(y + r)], r = 2 * scale[octave], covers it.  One workgroup per pair; rows with more than kRowCap entries
// are marked overflowed (count > kRowCap) and k_stereo_match scans all right keypoints for them, as it did for every row in r01.
__global__ __launch_bounds__(256) void k_stereo_rows(const Config* __restrict__ cfg, StereoArgs A, int inLds)
{
    extern __shared__ int s_rows[];                    // inLds: counts [H], then the lists [H][kRowCap] as u16
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int H = cfg->lv[0].h;
    const int nR = A.cntR[pair * A.cntStride];
    int* cnt = A.rowCnt + (size_t)pair * H;
    unsigned short* list = A.rowList + (size_t)pair * H * kRowCap;
    const ivf_keypoint* kpR = A.kpR + (size_t)pair * A.kpStride;
    __shared__ float s_scaleW[4][kMaxLevels];
    float* sc8 = s_scaleW[threadIdx.x >> 6];
    stage_scales(cfg, sc8, threadIdx.x & 63);
    // r06: four right keypoints per thread in flight (y and octave only), instead of one record per iteration with its use behind it
    auto for_each_right = [&](auto&& f) {
        for (int i0 = tid; i0 < nR; i0 += 4 * 256) {
            float ky[4]; int ko[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const ivf_keypoint* k = kpR + min(i0 + 256 * j, nR - 1); ky[j] = k->y; ko[j] = k->octave; }
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(ky[j]), "+v"(ko[j]));
#pragma unroll
            for (int j = 0; j < 4; j++) if (i0 + 256 * j < nR) f(i0 + 256 * j, ky[j], ko[j]);
        }
    };
    if (inLds) {
        // r04: the table of one pair (H x (4 + 2 kRowCap) bytes: 73 KB at H = 375) is built in LDS -- ~5 LDS atomics per right keypoint
        // instead of global ones (39 -> 13 us per 128 pairs) -- and leaves as one pass of coalesced dword stores; the order of a row's
        // entries is whatever the atomics give, as before (the matcher takes the minimum of (distance, index))
        int* cs = s_rows;
        unsigned short* ls = (unsigned short*)(s_rows + ((H + 3) & ~3));
        for (int y = tid; y < H; y += 256) cs[y] = 0;
        __syncthreads();
        for_each_right([&](int iR, float ky, int koct) {
            const float r = 2.0f * sc8[koct];
            const int maxr = min((int)ceilf(ky + r), H - 1), minr = max((int)floorf(ky - r), 0);
            for (int yi = minr; yi <= maxr; yi++) {
                const int pos = atomicAdd(&cs[yi], 1);
                if (pos < kRowCap) ls[yi * kRowCap + pos] = (unsigned short)iR;
            }
        });
        __syncthreads();
        for (int y = tid; y < H; y += 256) cnt[y] = cs[y];
        static_assert(kRowCap % 2 == 0, "rows are copied as dwords");
        const unsigned* ld = (const unsigned*)ls;
        unsigned* gd = (unsigned*)list;
        for (int i = tid; i < H * (kRowCap / 2); i += 256) {
            const int y = i / (kRowCap / 2), q = i % (kRowCap / 2);
            if (2 * q < min(cs[y], kRowCap)) gd[i] = ld[i];           // the second half of the last dword may be stale: never read (count)
        }
        return;
    }
    for (int y = tid; y < H; y += 256) cnt[y] = 0;
    __syncthreads();
    for_each_right([&](int iR, float ky, int koct) {
        const float r = 2.0f * sc8[koct];
        const int maxr = min((int)ceilf(ky + r), H - 1), minr = max((int)floorf(ky - r), 0);
        for (int yi = minr; yi <= maxr; yi++) {
            const int pos = atomicAdd(&cnt[yi], 1);
            if (pos < kRowCap) list[(size_t)yi * kRowCap + pos] = (unsigned short)iR;
        }
    });
}

__global__ __launch_bounds__(256) void k_stereo_match(const Config* __restrict__ cfg, StereoArgs A)
{
    const int pair = blockIdx.y;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nL = A.cntL[pair * A.cntStride], nR = A.cntR[pair * A.cntStride];
    if (iL >= cfg->nfeatures) return;
    float* uright = A.uright + (size_t)pair * A.outStride;
    float* depth = A.depth + (size_t)pair * A.outStride;
    int* sad = A.sad + (size_t)pair * A.outStride;
    if (iL >= nL) { if (lane == 0) { uright[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1; } return; }
    const ivf_keypoint* kpL = A.kpL + (size_t)pair * A.kpStride;
    const ivf_keypoint* kpR = A.kpR + (size_t)pair * A.kpStride;
    const uint8_t* descL = A.descL + (size_t)pair * A.kpStride * 32;
    const uint8_t* descR = A.descR + (size_t)pair * A.kpStride * 32;
    float outU = -1.0f, outD = -1.0f; int outS = -1;

    // r04: the kernel is a chain of dependent memory round trips at one wave per keypoint (SQ counters: waves waiting 82 % of their
    // cycles).  Everything whose ADDRESS is known after the left keypoint has arrived is requested at once -- the row's count, the
    // row's first 64 list entries, the left SAD window -- a candidate's descriptor is requested together with its keypoint record
    // (not after the band / octave / disparity tests), and the winner's x travels with the minimum instead of being read back:
    // 4 round trips (keypoint -> row list -> candidates -> right windows) instead of 8.
    // r06: the pyramid scales in the wave's LDS row up front (stage_scales), and the keypoint's level as a scalar: `cfg->lv[kl.octave]` and
    // `cfg->scale[kr.octave]` behind loaded octaves were vector loads -- two more dependent round trips in a kernel that is a chain of them
    const ivf_keypoint kl = kpL[iL];
    __shared__ float s_scaleW[4][kMaxLevels];
    float* sc8 = s_scaleW[threadIdx.x >> 6];
    stage_scales(cfg, sc8, lane);                                     // (scalar loads + four LDS writes of one lane, under the keypoint's round trip)
    const int levelL = __builtin_amdgcn_readfirstlane(kl.octave);      // one wave = one left keypoint: uniform
    const float vL = kl.y, uL = kl.x;
    const int row = (int)vL;
    const float minZ = A.bb, minD = 0, maxD = A.bf / minZ;
    const float minU = uL - maxD, maxU = uL - minD;
    const LevelGeom& G0 = cfg->lv[0];
    const bool live = row >= 0 && row < G0.h && !(maxU < 0);
    // left SAD window (depends on the left keypoint only)
    const float scaleFactor = cfg->invScale[levelL];
    const float scaleduL = roundf(kl.x * scaleFactor), scaledvL = roundf(kl.y * scaleFactor);
    const int w = 5, L = 5;
    const LevelGeom& G = cfg->lv[levelL];
    const uint8_t* PL = A.pyrL + (size_t)pair * A.pyrStride + G.off;
    const uint8_t* PR = A.pyrR + (size_t)pair * A.pyrStride + G.off;
    const int yl = (int)(scaledvL - w), xl = (int)(scaleduL - w);
    const bool okL = live && yl >= 0 && yl + 11 <= G.h && xl >= 0 && xl + 11 <= G.w;
    // r04: both SAD windows travel as whole dwords into LDS -- the left one (11 rows x 4 dwords from the 4-byte boundary left of xl) now, with
    // the keypoint; the right one (11 rows x 6 dwords: 21 pixels = 11 shifts x 11 columns) once the match is known -- one or two coalesced
    // loads per lane instead of 2 + 33 single-byte gathers per keypoint; the 121 x 11 absolute differences then read bytes from LDS
    __shared__ __attribute__((aligned(16))) unsigned s_win[4][128];        // per wave: [0, 44) left rows, [48, 114) right rows
    unsigned* const winL = s_win[threadIdx.x >> 6];
    unsigned* const winR = winL + 48;
    const int p0y = lane / 11, p0x = lane % 11, p1y = (lane + 64) / 11, p1x = (lane + 64) % 11;
    const bool has1 = lane + 64 < 121;
    unsigned wl = 0u;
    if (okL && lane < 44) wl = *(const unsigned*)(PL + (size_t)(yl + lane / 4) * G.pitch + (xl & ~3) + 4 * (lane % 4));
    unsigned best = (100u << 16) | 0xffffu;            // (dist << 16) | iR ; TH_HIGH start, strict <
    float bestX = 0.0f;                                // kpR[iR].x of this lane's best
    if (live) {
        const uint4* dl = (const uint4*)(descL + (size_t)iL * 32);
        const uint4 l0 = dl[0], l1 = dl[1];
        // candidates: the row's list (vRowIndices[vL], Frame.cc:806) when it did not overflow, every right keypoint otherwise;
        // the band / octave / disparity tests below are the reference's and make both enumerations equivalent
        const unsigned short* rl = A.rowCnt ? A.rowList + ((size_t)pair * G0.h + row) * kRowCap : nullptr;
        const int nRow = A.rowCnt ? A.rowCnt[(size_t)pair * G0.h + row] : kRowCap + 1;
        const int first = rl ? (int)rl[lane] : 0;      // lane < 64 <= kRowCap: inside the row's slot whatever the count is
        const bool listed = nRow <= kRowCap;
        const int nCand = listed ? nRow : nR;
        for (int base = 0; base < nCand; base += 64) {
            const int ci = base + lane;
            const int iR = ci < nCand ? (listed ? (base == 0 ? first : (int)rl[ci]) : ci) : nR;
            if (iR < nR) {
                const ivf_keypoint kr = kpR[iR];
                const uint4* dr = (const uint4*)(descR + (size_t)iR * 32);
                const uint4 r0 = dr[0], r1 = dr[1];
                const float r = 2.0f * sc8[kr.octave];
                const int maxr = (int)ceilf(kr.y + r), minr = (int)floorf(kr.y - r);
                if (row >= minr && row <= maxr && kr.octave >= levelL - 1 && kr.octave <= levelL + 1 &&
                    kr.x >= minU && kr.x <= maxU) {
                    const unsigned d = (unsigned)hamming256(l0, l1, r0, r1);
                    const unsigned key = (d << 16) | (unsigned)iR;
                    if (d < 100u && key < best) { best = key; bestX = kr.x; }
                }
            }
        }
    }
    const unsigned mine = best;
    best = wave_min_u32(best);
    const int bestDist = best >> 16;
    const int bestIdxR = best & 0xffff;
    if (live && bestIdxR != 0xffff && bestDist < 75) {
        // the key holds iR, so exactly the lanes that found this candidate hold it (one lane: a candidate is enumerated once)
        const unsigned long long who = __ballot(mine == best);
        const float uR0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bestX), __ffsll((long long)who) - 1));
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        const bool ok = !(iniu < 0 || endu >= (float)G.w) && okL && (int)(scaleduR0 - L - w) >= 0;
        if (ok) {
            const int x0R = (int)(scaleduR0 - (float)(L + w));                    // leftmost right-window column (>= 0: checked)
            unsigned wr0 = 0u, wr1 = 0u;                                          // 66 dwords: lane, lane + 64
            {
                const uint8_t* rowp = PR + (size_t)yl * G.pitch + (x0R & ~3);
                wr0 = *(const unsigned*)(rowp + (size_t)(lane / 6) * G.pitch + 4 * (lane % 6));
                if (lane < 2) wr1 = *(const unsigned*)(rowp + (size_t)((lane + 64) / 6) * G.pitch + 4 * ((lane + 64) % 6));
            }
            if (lane < 44) winL[lane] = wl;
            winR[lane] = wr0;
            if (lane < 2) winR[lane + 64] = wr1;
            wave_sync_lds();
            const uint8_t* bL = (const uint8_t*)winL + (xl & 3);                  // row pitch 16 bytes
            const uint8_t* bR = (const uint8_t*)winR + (x0R & 3);                 // row pitch 24 bytes; shift inc = column offset inc + 5
            const int cL = bL[w * 16 + w];
            const int dl0 = (int)bL[p0y * 16 + p0x] - cL;
            const int dl1 = has1 ? (int)bL[p1y * 16 + p1x] - cL : 0;
            int bestD = 0x7fffffff, bestinc = 0;
            float vDists[11];
#pragma unroll
            for (int inc = -5; inc <= 5; inc++) {
                const int o = inc + L;                                            // (int)(scaleduR0 + inc - w) - x0R: the values are integers
                const int cR = bR[w * 24 + o + w];
                int acc = abs(dl0 - ((int)bR[p0y * 24 + o + p0x] - cR));
                if (has1) acc += abs(dl1 - ((int)bR[p1y * 24 + o + p1x] - cR));
                acc = wave_sum_i32(acc);
                const float dist = (float)acc;
                if (dist < (float)bestD) { bestD = (int)dist; bestinc = inc; }
                vDists[inc + 5] = dist;
            }
            if (!(bestinc == -L || bestinc == L)) {
                float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                for (int q = 1; q < 10; q++) if (q == bestinc + 5) { dist1 = vDists[q - 1]; dist2 = vDists[q]; dist3 = vDists[q + 1]; }
                const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = cfg->scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
                    float disparity = (uL - bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }     // Frame.cc:909: uL-0.01 in double, then narrowed
                        outD = A.bf / disparity; outU = bestuR; outS = bestD;
                    }
                }
            }
        }
    }
    if (lane == 0) { uright[iL] = outU; depth[iL] = outD; sad[iL] = outS; }
}

// k_stereo_gate: one workgroup = one pair.  sort(vDistIdx); median = vDistIdx[size/2].first;
// drop every match with dist >= 1.5f*1.4f*median (Frame.cc:918-931).  Only the median VALUE matters,
// so it is found by rank counting instead of sorting.  Empty list => no gate (Appendix D-8).
__global__ __launch_bounds__(256) void k_stereo_gate(const Config* __restrict__ cfg, const int* __restrict__ cntL,
                                                    int cntStride, float* __restrict__ uright, float* __restrict__ depth,
                                                    const int* __restrict__ sad, int outStride)
{
    // SAD distances are integers in [0, 121*510]: the value at sorted index n/2 comes from two 256-bin histograms
    // (high byte, then low byte inside the selected bin) instead of a sort.
    __shared__ int s_hist[256];
    __shared__ int s_n, s_bin, s_below, s_median;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nL = cntL[pair * cntStride];
    const int* S = sad + (size_t)pair * outStride;
    s_hist[tid] = 0;
    if (tid == 0) { s_n = 0; s_median = -1; }
    __syncthreads();
    int local = 0;
    // r06: four distances per thread in flight in each of the three passes (one load per iteration with its use behind it = one round trip per iteration)
    auto for_each_dist = [&](auto&& f) {
        for (int i0 = tid; i0 < nL; i0 += 4 * 256) {
            int dv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) dv[j] = S[min(i0 + 256 * j, nL - 1)];
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(dv[j]));
#pragma unroll
            for (int j = 0; j < 4; j++) if (i0 + 256 * j < nL) f(i0 + 256 * j, dv[j]);
        }
    };
    for_each_dist([&](int, int d) { if (d >= 0) { local++; atomicAdd(&s_hist[min(d >> 8, 255)], 1); } });
    if (local) atomicAdd(&s_n, local);
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    const int target = n / 2;
    // r06: the first bin whose running count exceeds the target, by a 256-thread inclusive scan (8 steps) instead of thread 0 walking up to 256 LDS reads twice
    // (most of this kernel's 17 us).  Exactly one thread sees incl > t >= excl; it is the bin the sequential walk stops at and excl is what the walk had summed.
    __shared__ int s_scan[256];
    auto first_exceed = [&](int t, int& bin, int& below) {
        const int v = s_hist[tid];
        s_scan[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int a = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += a;
            __syncthreads();
        }
        const int incl = s_scan[tid], excl = incl - v;
        if (incl > t && excl <= t) { s_bin = tid; s_below = excl; }
        __syncthreads();
        bin = s_bin; below = s_below;
    };
    int bin, below;
    first_exceed(target, bin, below);
    __syncthreads();
    s_hist[tid] = 0;
    __syncthreads();
    for_each_dist([&](int, int d) { if (d >= 0 && min(d >> 8, 255) == bin) atomicAdd(&s_hist[d & 255], 1); });
    __syncthreads();
    {
        int b2, below2;
        first_exceed(target - below, b2, below2);
        if (tid == 0) s_median = (bin << 8) | b2;
    }
    __syncthreads();
    const float median = (float)s_median;
    const float thDist = 1.5f * 1.4f * median;
    for_each_dist([&](int i, int d) {
        if (d >= 0 && !((float)d < thDist)) {
            uright[(size_t)pair * outStride + i] = -1;
            depth[(size_t)pair * outStride + i] = -1;
        }
    });
}

// testing hook: the device's retainBest on caller-supplied responses (one wave, LDS), so the wave-cooperative
// introselect can be checked against libstdc++ on arbitrary / adversarial inputs
__global__ __launch_bounds__(64) void k_test_retain_best(const float* __restrict__ resp, int n, int nPoints, int* __restrict__ order)
{
    __shared__ __attribute__((aligned(16))) u64 list[kCellCapBig];
    __shared__ unsigned short sa[kCellCapBig], sb[kCellCapBig];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) list[i] = ((u64)__float_as_uint(resp[i]) << 32) | (unsigned)i;
    wave_sync_lds();
    if (nPoints > 0 && n > nPoints) sel_nth_element_wave(list, n, nPoints - 1, sa, sb, lane);
    for (int i = lane; i < n; i += 64) order[i] = (int)(unsigned)list[i];
}
void launch_test_retain_best(const float* dResp, int n, int nPoints, int* dOrder, hipStream_t s)
{
    hipLaunchKernelGGL(k_test_retain_best, dim3(1), dim3(64), 0, s, dResp, n, nPoints, dOrder);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// dstBlob == b.qpyr: the cost images; only the planes something reads are written (useCost bit 0: the cost pyramid gates the
// extraction, bit 1: mvKeyQualScore samples level 0)
void launch_ingest(const Config& hc, const Config* dc, const Buffers& b, const uint8_t* src0, const uint8_t* src1,
                   size_t imageStride, int rowStride, int nImg, int nSides, uint8_t* dstBlob, hipStream_t s, int sideMask)
{
    const LevelGeom& G = hc.lv[0];
    dim3 grid((G.h + kIngestRows - 1) / kIngestRows, nImg);
    hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, s, dc, src0, src1, imageStride, rowStride, nSides, dstBlob,
                       dstBlob == b.qpyr && b.qpyr ? b.useCost : (const uint8_t*)nullptr, sideMask);
}
// side `side` of every pair (nImg / nSides images) from 3-channel interleaved images: level 0 = cvtColor(..., GRAY)
void launch_ingest_color(const Config& hc, const Config* dc, const uint8_t* src, size_t imageStride, int rowStride, int code, int nImg, int nSides, int side,
                         uint8_t* dstBlob, hipStream_t s)
{
    const LevelGeom& G = hc.lv[0];
    dim3 grid((G.h + kIngestRows - 1) / kIngestRows, nImg / nSides);
    hipLaunchKernelGGL(k_ingest_color, grid, dim3(256), 0, s, dc, src, imageStride, rowStride, code, nSides, side, dstBlob);
}
// one launch per level: planes of `blob` and (qblob != nullptr) of the cost blob, the latter only for images whose useCost bit 0 is set
void launch_pyramid(const Config& hc, const Config* dc, const ResizeCoef* dTab, uint8_t* blob, uint8_t* qblob, const uint8_t* useCost,
                    int nImg, hipStream_t s)
{
    static const bool ldsOk = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_down), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  kPyrMaxR * kPyrMaxP) == hipSuccess;
    (void)ldsOk;
    // LDS tile of a level: the source rows / bytes a 256 x 32 tile can draw from at this level ratio (+ alignment slack); a
    // ratio above 2 does not fit kPyrMaxR x kPyrMaxP and takes the kernel's un-staged path
    auto tile_rows = [&](int l) { return std::min(kPyrMaxR, (int)((double)kPyrTH * hc.lv[l - 1].h / hc.lv[l].h) + 4); };
    auto tile_pitch = [&](int l) { return std::min(kPyrMaxP, (((int)((double)kPyrTW * hc.lv[l - 1].w / hc.lv[l].w) + 2 + 15 + 16) / 16) * 16); };
    // r04 measured the upper levels merged into ONE launch (k_pyr_multi, a counter barrier among the workgroups of a plane between
    // levels): levels 4-7 merged 83 us against 75 us for their four launches -- not faster, and its barrier had open defects
    // (r04 ADVICE): removed in r05; DESIGN.md section 5.r04 keeps the measurement.
    for (int l = 1; l < hc.nlevels; l++) {
        const LevelGeom& G = hc.lv[l];
        if (G.w <= 0 || G.h <= 0) continue;
        const int rows = tile_rows(l), pitch = tile_pitch(l);
        dim3 grid((G.pitch + kPyrTW - 1) / kPyrTW, (G.h + kPyrTH - 1) / kPyrTH, qblob ? 2 * nImg : nImg);
        hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), (size_t)rows * pitch, s, dc, l, dTab, blob, qblob, useCost, nImg, pitch, rows);
    }
}
void launch_fast(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s)
{
    if (hc.nTiles <= 0) return;
    hipLaunchKernelGGL(k_fast_nms, dim3((nImg + 7) / 8 * 8 * hc.nTiles), dim3(256), 0, s, dc, b.pyr, b.useCost, b.tileList, b.tileCnt, b.cellCnt,
                       nImg);
}
void launch_blur(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s, bool skipEmptyLevels)
{
    const int tiles = hc.nBlurTiles;
    if (tiles <= 0) return;
    // skipEmptyLevels reads lvlCount, which k_level_select writes: only when the blur runs behind the selection
    hipLaunchKernelGGL(k_blur7, dim3((nImg + 7) / 8 * 8 * tiles), dim3(256), 0, s, dc, b.pyr, skipEmptyLevels ? b.lvlCount : nullptr, b.blur, nImg);
}
void launch_select(const Config& hc, const Config* dc, const Buffers& b, int nImg, hipStream_t s)
{
    if (hc.introspection)
        hipLaunchKernelGGL(k_cell_qsum, dim3((hc.nCellsTotal + 3) / 4, nImg), dim3(256), 0, s, dc, b.qpyr, b.useCost, (CellInfo*)b.cellInfo);
    const int tierCap = nImg * hc.nCellsTotal;
    int cap = 4;
    for (int l = 0; l < hc.nlevels; l++) cap = std::max(cap, (hc.lv[l].nCells + 3) & ~3);
    const size_t quotaLds = (size_t)cap * (7 * 4 + 1) + 64;                     // six int / float arrays + the prefix's tail + the byte flags (k_quota)
    hipLaunchKernelGGL(k_quota, dim3(hc.nlevels, nImg), dim3(256), quotaLds, s, dc, b.cellCnt, b.qpyr, b.useCost,
                       (CellInfo*)b.cellInfo, b.lvlTotal, b.hugeCount, b.hugeList, b.tierList, tierCap, b.status, cap);
    // three tiers by survivor count: the kernel is a chain of dependent LDS steps at one wave per cell, so its throughput is the
    // number of cells in flight per CU = LDS per workgroup: 5 KB (<= 256 survivors), 14 KB (<= 1024), 51 KB (<= 4096, four waves; a 2048 tier of two waves measured slower: 77 + 42 vs 109 us)
    // r04: tier 0 four cells per workgroup; tiers 1 / 2 from k_quota's work lists with a fixed grid (a full grid per tier cost 80-110 us
    // of workgroup dispatch per launch even when the tier was empty, as it is for every cell of a plain ORB extraction)
    constexpr int kSelWPB = 4;
    hipLaunchKernelGGL((k_cell_select<kCellCapTiny, kSelWPB>), dim3((hc.nCellsTotal + kSelWPB - 1) / kSelWPB, nImg), dim3(64 * kSelWPB), 0, s, dc,
                       b.tileList, b.tileCnt, (const CellInfo*)b.cellInfo, b.qpyr, b.useCost, b.lvl, b.status, 0, kCellCapTiny);
    const int gridList = std::max(1, std::min(tierCap, 8192));
    hipLaunchKernelGGL((k_cell_select_list<kCellCapSmall, 64>), dim3(gridList), dim3(64), 0, s, dc, b.tileList, b.tileCnt,
                       (const CellInfo*)b.cellInfo, b.qpyr, b.useCost, b.lvl, b.status, kCellCapTiny, kCellCapSmall, b.hugeCount + 1, b.tierList, tierCap);
    hipLaunchKernelGGL((k_cell_select_list<kCellCapBig, 256>), dim3(std::min(gridList, 2048)), dim3(256), 0, s, dc, b.tileList, b.tileCnt,
                       (const CellInfo*)b.cellInfo, b.qpyr, b.useCost, b.lvl, b.status, kCellCapSmall, kCellCapBig, b.hugeCount + 2,
                       b.tierList + tierCap, tierCap);
    if (b.hugeScratch)       // geometry allows cells with more than kCellCapBig strict maxima: walk k_quota's (usually empty) list
        hipLaunchKernelGGL(k_cell_select_huge, dim3(kHugeSlots), dim3(1024), 0, s, dc, b.tileList, b.tileCnt, (const CellInfo*)b.cellInfo,
                           b.qpyr, b.useCost, b.lvl, b.status, b.hugeCount, b.hugeList, b.hugeScratch, hc.maxCandCap);
    int maxDesired = 1;
    for (int l = 0; l < hc.nlevels; l++) maxDesired = std::max(maxDesired, hc.lv[l].nDesired);
    const int lcap = std::min(4096, (4 * maxDesired + 256 + 63) & ~63);
    hipLaunchKernelGGL(k_level_select, dim3(hc.nlevels, nImg), dim3(256), (size_t)lcap * 12, s, dc, b.lvlTotal, b.lvl, b.slotPos, b.slotResp,
                       b.lvlCount, lcap);
}
void launch_describe(const Config& hc, const Config* dc, const Buffers& b, const uint8_t*, size_t, int, int nImg, int,
                     hipStream_t s)
{
    hipLaunchKernelGGL(k_describe, dim3((nImg + 7) / 8 * 8 * ((hc.nfeatures + kDescKP - 1) / kDescKP)), dim3(256), 0, s, dc, b.pyr, b.blur, b.qpyr,
                       b.useCost, b.slotPos, b.slotResp, b.lvlCount, b.kps, b.desc, b.count, b.quality, nImg);
}
void launch_stereo_args(const Config& hc, const Config* dc, const StereoArgs& A, int nPairs, hipStream_t s)
{
    if (A.rowCnt) {
        // the row table in LDS when it fits (the count array is padded to a multiple of 4 ints so that the lists start dword-aligned)
        const size_t lds = (size_t)((hc.lv[0].h + 3) & ~3) * 4 + (size_t)hc.lv[0].h * kRowCap * 2;
        static const bool big = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stereo_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    150 * 1024) == hipSuccess;
        const bool inLds = big && lds <= 150 * 1024;
        hipLaunchKernelGGL(k_stereo_rows, dim3(nPairs), dim3(256), inLds ? lds : 0, s, dc, A, inLds ? 1 : 0);
    }
    hipLaunchKernelGGL(k_stereo_match, dim3((hc.nfeatures + 3) / 4, nPairs), dim3(256), 0, s, dc, A);
    hipLaunchKernelGGL(k_stereo_gate, dim3(nPairs), dim3(256), 0, s, dc, A.cntL, A.cntStride, A.uright, A.depth, A.sad,
                       A.outStride);
}
void launch_stereo(const Config& hc, const Config* dc, const Buffers& b, int nPairs, float bf, float bb, hipStream_t s)
{
    StereoArgs A;
    A.pyrL = b.pyr; A.pyrR = b.pyr + hc.pyrBytes; A.pyrStride = (size_t)2 * hc.pyrBytes;
    A.kpL = b.kps; A.kpR = b.kps + hc.nfeatures; A.descL = b.desc; A.descR = b.desc + (size_t)hc.nfeatures * 32;
    A.cntL = b.count; A.cntR = b.count + 1; A.kpStride = (size_t)2 * hc.nfeatures; A.cntStride = 2;
    A.uright = b.uright; A.depth = b.depth; A.sad = b.sad; A.outStride = hc.nfeatures;
    A.bf = bf; A.bb = bb; A.rowCnt = b.rowCnt; A.rowList = b.rowList;
    launch_stereo_args(hc, dc, A, nPairs, s);
}
// ---- MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:281-305): per observed descriptor the median of
// its Hamming distances to all n (the 0 of the diagonal included) = sorted row [(int)(0.5*(n-1))].  Distances live in
// 0..256, so the median is read off a 257-bin LDS histogram instead of a sort: the smallest value whose cumulative count
// exceeds the index.  One workgroup per row.
__global__ __launch_bounds__(256) void k_distinct_median(const uint8_t* __restrict__ desc, int n, int* __restrict__ median)
{
    __shared__ int hist[257 + 7];
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < 264; k += 256) hist[k] = 0;
    __syncthreads();
    const uint4* pi = (const uint4*)(desc + (size_t)i * 32);
    const uint4 a0 = pi[0], a1 = pi[1];
    for (int j = tid; j < n; j += 256) {
        const uint4* pj = (const uint4*)(desc + (size_t)j * 32);
        const int d = j == i ? 0 : hamming256(a0, a1, pj[0], pj[1]);
        atomicAdd(&hist[d], 1);
    }
    __syncthreads();
    if (tid == 0) {
        const int target = (int)(0.5 * (n - 1));
        int acc = 0, v = 0;
        for (; v < 257; v++) { acc += hist[v]; if (acc > target) break; }
        median[i] = v;
    }
}

void launch_distinct_median(const uint8_t* desc, int n, int* median, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_distinct_median, dim3(n), dim3(256), 0, s, desc, n, median);
}

// ---- DBoW2 vocabulary-tree descent (TemplatedVocabulary.h:1217-1259): 16 lanes per descriptor, one child per lane per
// round, (distance << 16 | position) min-reduction across the 16 lanes = "first minimum in child order"
__global__ __launch_bounds__(256) void k_bow_transform(const int* __restrict__ childStart, const int* __restrict__ child,
                                                      const uint8_t* __restrict__ nodeDesc, const uint8_t* __restrict__ desc,
                                                      int n, int nidLevel, int* __restrict__ leaf, int* __restrict__ nodeAt)
{
    const int f = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    const bool live = f < n;
    const uint4* pf = (const uint4*)(desc + (size_t)(live ? f : 0) * 32);
    const uint4 a0 = pf[0], a1 = pf[1];
    int node = 0, level = 0, nid = 0;
    for (;;) {
        const int c0 = childStart[node], c1 = childStart[node + 1];
        if (c1 == c0) break;                                              // leaf (uniform within the 16 lanes)
        ++level;
        unsigned best = 0xffffffffu;
        for (int c = c0 + sub; c < c1; c += 16) {
            const int id = child[c];
            const uint4* pn = (const uint4*)(nodeDesc + (size_t)id * 32);
            const unsigned key = ((unsigned)hamming256(a0, a1, pn[0], pn[1]) << 16) | (unsigned)(c - c0);
            best = min(best, key);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o, 16));
        node = child[c0 + (int)(best & 0xffffu)];
        if (level == nidLevel) nid = node;
    }
    if (live && sub == 0) { leaf[f] = node; nodeAt[f] = nidLevel <= 0 ? 0 : nid; }
}

void launch_bow_transform(const int* childStart, const int* child, const uint8_t* nodeDesc, const uint8_t* desc, int n, int nidLevel,
                          int* leaf, int* nodeAt, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bow_transform, dim3((n * 16 + 255) / 256), dim3(256), 0, s, childStart, child, nodeDesc, desc, n, nidLevel, leaf, nodeAt);
}

void launch_hamming_pairs(const uint8_t* a, const uint8_t* b, const int* pairs, int n, int* dist, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, s, a, b, pairs, n, dist);
}

// ------------------------------------------------------------------------------------------------
// k_pack_gather: one workgroup = one stereo pair; copies the left frame's {count, keypoints, descriptors, uRight} into
// one contiguous record {n, kps, desc, uRight, depth} of the all-gather block (depth so that a receiving rank can
// un-project the frame's stereo points, ivf_track.hip).  A kernel rather than 2-D copies so that packing
// never blocks the host thread that keeps the next batches enqueued.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_gather(const int* __restrict__ count, const ivf_keypoint* __restrict__ kps,
                                                     const uint8_t* __restrict__ desc, const float* __restrict__ uright,
                                                     const float* __restrict__ depth, int nf, unsigned* __restrict__ block, size_t recWords)
{
    const int p = blockIdx.x;
    unsigned* out = block + (size_t)p * recWords;
    if (threadIdx.x < 4) out[threadIdx.x] = threadIdx.x == 0 ? (unsigned)count[2 * p] : 0u;
    const unsigned* k = (const unsigned*)(kps + (size_t)2 * p * nf);
    const unsigned* d = (const unsigned*)(desc + (size_t)2 * p * nf * 32);
    const unsigned* u = (const unsigned*)(uright + (size_t)p * nf);
    const unsigned* z = (const unsigned*)(depth + (size_t)p * nf);
    const int nk = nf * (int)(sizeof(ivf_keypoint) / 4), nd = nf * 8;
    // r06: eight dwords per thread in flight (a plain `out[i] = in[i]` loop is one round trip per iteration: 64 of them per pair at N = 1000)
    auto copy = [&](unsigned* __restrict__ o, const unsigned* __restrict__ in, int n) {
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
            unsigned v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = in[min(i0 + 256 * j, n - 1)];
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" : "+v"(v[j]));
#pragma unroll
            for (int j = 0; j < 8; j++) if (i0 + 256 * j < n) o[i0 + 256 * j] = v[j];
        }
    };
    copy(out + 4, k, nk);
    copy(out + 4 + nk, d, nd);
    copy(out + 4 + nk + nd, u, nf);
    copy(out + 4 + nk + nd + nf, z, nf);
}

void launch_pack_gather(const Buffers& b, int nf, int nPairs, uint8_t* block, size_t recBytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_gather, dim3(nPairs), dim3(256), 0, s, b.count, b.kps, b.desc, b.uright, b.depth, nf, (unsigned*)block, recBytes / 4);
}

// ------------------------------------------------------------------------------------------------
// Device-resident frame grid (Frame::AssignFeaturesToGrid / GetFeaturesInArea, ORB/src/Frame.cc:415-430, 615-680).
// k_grid_build: ONE workgroup builds the 64x48 bucket grid of a frame in CSR form with the buckets in the reference's
//   enumeration order (cell = ix * 48 + iy) and the keypoints of a bucket in insertion order (a stable counting sort:
//   the rank of a keypoint inside its bucket = keypoints of the same bucket in earlier 256-blocks + earlier threads of
//   its own block).  start: [64*48 + 1], idx: [n].
// k_grid_window: one wave per query.  The buckets of one grid column are contiguous in `idx`, so the window is at most
//   64 contiguous runs; every run is walked 64 candidates at a time: octave / |dx| < r / |dy| < r filters as the
//   reference applies them (:636-664), Hamming distance of the survivors against the query descriptor, and an ordered,
//   ballot-compacted append of (index, distance) to the query's candidate list (cap entries; count may exceed cap =
//   overflow, the host then re-does that query through the pair path).
// ------------------------------------------------------------------------------------------------
constexpr int kGC = 64, kGR = 48;
__global__ __launch_bounds__(256) void k_grid_build(const ivf_keypoint* __restrict__ kps, int n, float minX, float minY,
                                                   float invW, float invH, int* __restrict__ start, int* __restrict__ idx)
{
    __shared__ int cnt[kGC * kGR];
    __shared__ int part[256];
    __shared__ int blk[256];
    const int tid = threadIdx.x;
    for (int c = tid; c < kGC * kGR; c += 256) cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](int i) {
        const int px = (int)roundf((kps[i].x - minX) * invW), py = (int)roundf((kps[i].y - minY) * invH);   // Frame::PosInGrid :672-673
        return (px < 0 || px >= kGC || py < 0 || py >= kGR) ? -1 : px * kGR + py;
    };
    for (int i = tid; i < n; i += 256) { const int c = cell_of(i); if (c >= 0) atomicAdd(&cnt[c], 1); }
    __syncthreads();
    // exclusive prefix over the 3072 buckets: 12 per thread + a block scan of the partial sums
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
    for (int k = 0; k < PER; k++) { start[tid * PER + k] = base + local[k]; cnt[tid * PER + k] = base + local[k]; }   // cnt becomes the fill cursor
    if (tid == 255) start[kGC * kGR] = part[255];
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const int c = i < n ? cell_of(i) : -1;
        blk[tid] = c;
        __syncthreads();
        if (c >= 0) {
            int before = 0;
            for (int t = 0; t < tid; t++) before += blk[t] == c ? 1 : 0;
            idx[cnt[c] + before] = i;
        }
        __syncthreads();
        if (c >= 0) atomicAdd(&cnt[c], 1);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_grid_window(const ivf_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                    const int* __restrict__ start, const int* __restrict__ idx,
                                                    float minX, float minY, float invW, float invH, int nq,
                                                    const float* __restrict__ qu, const float* __restrict__ qv,
                                                    const float* __restrict__ qr, const int* __restrict__ qminL,
                                                    const int* __restrict__ qmaxL, const uint8_t* __restrict__ qdesc,
                                                    const uint8_t* __restrict__ qvalid, int cap, int* __restrict__ count,
                                                    int2* __restrict__ cand)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    int total = 0;
    if (!qvalid || qvalid[q]) {
        const float x = qu[q], y = qv[q], r = qr[q];
        const int minL = qminL[q], maxL = qmaxL[q];
        // Frame::GetFeaturesInArea :620-634
        const int x0 = max(0, (int)floorf((x - minX - r) * invW)), x1 = min(kGC - 1, (int)ceilf((x - minX + r) * invW));
        const int y0 = max(0, (int)floorf((y - minY - r) * invH)), y1 = min(kGR - 1, (int)ceilf((y - minY + r) * invH));
        if (x0 < kGC && x1 >= 0 && y0 < kGR && y1 >= 0) {
            const bool chk = (minL > 0) || (maxL >= 0);
            const uint4* qd = (const uint4*)(qdesc + (size_t)q * 32);
            const uint4 qa = qd[0], qb = qd[1];
            for (int ix = x0; ix <= x1; ix++) {
                const int s = start[ix * kGR + y0], e = start[ix * kGR + y1 + 1];      // buckets iy = y0..y1 are contiguous
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
                            const uint4 a = cd[0], b2 = cd[1];
                            d = __popc(a.x ^ qa.x) + __popc(a.y ^ qa.y) + __popc(a.z ^ qa.z) + __popc(a.w ^ qa.w) +
                                __popc(b2.x ^ qb.x) + __popc(b2.y ^ qb.y) + __popc(b2.z ^ qb.z) + __popc(b2.w ^ qb.w);
                        }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) {
                        const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
                        if (pos < cap) cand[(size_t)q * cap + pos] = make_int2(i2, d);
                    }
                    total += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) count[q] = total;
}

void launch_grid_build(const ivf_keypoint* kps, int n, float minX, float minY, float invW, float invH, int* start, int* idx, hipStream_t s)
{
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(256), 0, s, kps, n, minX, minY, invW, invH, start, idx);
}
void launch_grid_window(const ivf_keypoint* kps, const uint8_t* desc, const int* start, const int* idx, float minX, float minY,
                        float invW, float invH, int nq, const float* qu, const float* qv, const float* qr, const int* qminL,
                        const int* qmaxL, const uint8_t* qdesc, const uint8_t* qvalid, int cap, int* count, int* cand, hipStream_t s)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_grid_window, dim3((nq + 3) / 4), dim3(256), 0, s, kps, desc, start, idx, minX, minY, invW, invH, nq, qu, qv,
                       qr, qminL, qmaxL, qdesc, qvalid, cap, count, (int2*)cand);
}

}  // namespace ivf
