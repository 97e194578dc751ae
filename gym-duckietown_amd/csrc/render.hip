// render.hip -- Simulator.render_obs as a forward per-pixel raster for gfx950.
//
// Replaces simulator.py:1707-1951 (_render_img: fixed-function GL into a 4xMSAA FBO,
// glReadPixels, flip) and distortion.py:85-125 (cv2.remap INTER_NEAREST through the
// inverted rectify map).  Render spec: SURVEY.md Appendix B / DESIGN.md "Render spec";
// the CPU statement of the same spec is oracle/raster.py.
//
// Structure
//   k_cam_setup : one thread per env -> EnvCam (camera centre, yaw, colours, ground-corner
//                 lighting; with DR also pitch / frustum / light), 128 B per env.
//   k_raster<DR,OBJ>: workgroup = 4 independent wavefronts owning a DT_TILE_W x DT_TILE_H = 128 x 8 pixel
//                 tile of the frame (wavefront w: rows 2w, 2w+1; lane l: columns l and l+64 of each row), looping over
//                 ENVS_PER_BLOCK envs.  Adjacent lanes are adjacent pixels, so one texel-load
//                 instruction touches neighbouring texture lines (TA coalescing) and the tile's
//                 2-D texture footprint stays in L1.  Everything that does not depend on the env
//                 -- the LUT entry of each pixel (NDC of the rectilinear source pixel: the
//                 fisheye remap is folded into the ray set-up, there is no second pass) and, for
//                 the shared camera (DR=false), the whole ray / tile-plane intersection in the
//                 yaw-local frame, the light term and the MSAA edge margin -- is computed once
//                 and kept in REGISTERS across the env loop; per env only the 64-B EnvFast
//                 changes (one wave-uniform scalar load) and a pixel costs one 2-D rotation +
//                 tile lookup (LDS) + one bilinear fetch (two 8-byte loads from the L2-resident
//                 padded texture, llvmpipe's integer GL_LINEAR: gl_linear_rgb), times the lit factor.
//                 MSAA: pixels whose 4 samples may see different primitives (horizon, map
//                 border, tile seams, mesh boxes; conservative test) are appended (ballot +
//                 mbcnt, no atomics) to a per-wavefront global queue region.
//                 Output: px -> wavefront-private LDS transpose -> 12 B (4 px) per lane, three
//                 dword stores, 384 contiguous bytes per tile row.
//   k_blk_setup / k_obj_setup (maps with objects): source-pixel boxes of the raster wavefront blocks (env-invariant);
//                 one workgroup per env projects the env's mesh triangles to screen space (per-vertex lighting,
//                 texture index, traffic-light card by pattern, segmentation colour), reduces per-object screen boxes
//                 and writes, per (env, block), the mask of the objects whose box meets the block.
//   k_raster_q<OBJ,S256> (shared camera, square power-of-two tile textures): quad-record one-ray path +
//                 the exact path of plane-edge pixels inside the same wavefronts (resolve_region).
//   k_raster_v3<OBJ> (render_v3.inc; 256 x 256 tile textures, the BASELINE configurations): k_raster_q on an instruction diet -- the
//                 headline kernel; k_raster_v3dr<OBJ> (render_v3dr.inc): domain randomisation on the same records (per-env homography).
//   k_env_sort:   render order of the envs for the quad-record paths (tile under the camera, heading quadrant): an XCD's L2 serves one
//                 region of the map; DTSIM_FIELD_RENDER_POS reads the order back.
//   k_resolve:    exact 4-sample resolve of the plane-edge pixels the generic k_raster / k_raster_v3dr queued (front of the queue
//                 regions): persistent wavefronts pull work items (8 queue batches of one raster workgroup), 64 entries
//                 at a time -- coverage per sample, shading once per distinct primitive at the pixel centre --
//                 and patch the 3 bytes of each pixel; stream-ordered after the raster.
//   k_resolve_obj: the pixels inside mesh-object screen boxes (far end of the queue regions, either raster), one unit
//                 per (raster tile, env): triangles streamed and staged in LDS once per unit, z-buffered per sample in
//                 1/depth space, then the same shading.
//   Segmentation render (dtsim_render_ex): same kernels on the segmented texel pool, k_cam_setup forces the unlit
//                 state and the magenta clear / ground colour, k_obj_setup folds the mesh's flat colour into Kd.
//
// Roofline: algorithmic bytes per env-step = W*H*3 (921 600 B at 640x480), written once (+ the ~1.3 % edge
// pixels a second time); LUT / textures / tables are shared by all envs and stay in registers / LDS / L2.
// Measured (profiles/): the pass is co-limited by vector issue and the texture path's cache-line rate -- 31.1 vector instructions per
// pixel (round 1: 64.8; round 5: 36.9), 12 lines per record gather, vector ALUs 82 % and texture unit 73 % busy -- at 31.5 - 32.3 % of the HBM roofline; float32 geometry, byte-weight
// filter at the precision GL's own GL_LINEAR has on the reference's renderer, uint8 output.
// Parity: every pipeline is held to frames the unmodified reference rendered on Mesa llvmpipe (tests/test_gpu_gl_golden.py).
#include "dtsim_dev.h"
#include <hip/hip_fp16.h>
#include <type_traits>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

// Buffer stores through a raw descriptor (k_raster_v3 / k_raster_v3dr frame rows): clang has no builtin with a scalar-base form, the LLVM
// intrinsics are reached by name (their immediates must be literals after inlining).
typedef int v3_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t v3_u32x3 __attribute__((ext_vector_type(3)));
__device__ void v3_buf_store_v3i32(v3_u32x3 data, v3_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v3i32");
typedef uint32_t v3_u32x4 __attribute__((ext_vector_type(4)));
__device__ void v3_buf_store_v4i32(v3_u32x4 data, v3_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
#define V3_RSRC_W3 0x00027000      // raw buffer descriptor word 3
__device__ inline v3_i32x4 v3_rsrc(const void* p, uint32_t bytes) {
  v3_i32x4 r;
  r.x = (int)(uint32_t)reinterpret_cast<uintptr_t>(p); r.y = (int)((reinterpret_cast<uintptr_t>(p) >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = V3_RSRC_W3;
  return r;
}


#define RB 256            // threads per workgroup (4 wavefronts)
#define PPT DT_PPT         // pixels per thread
#define WAVE_W DT_WAVE_W  // pixel columns per wavefront block (dtsim_dev.h)
// pixel slot k of lane l is pixel number k*64 + l of the block, row-major: adjacent lanes = adjacent pixels
#define SLOT_X(k, l) (((k) * 64 + (l)) % WAVE_W)
#define SLOT_Y(k, l) (((k) * 64 + (l)) / WAVE_W)
#define WAVE_PIX (64 * PPT)
#define ENVS_PER_BLOCK DT_ENVS_PER_BLOCK
#ifndef DT_TRI_CAP
#define DT_TRI_CAP 96              // round 3: 128 -> 96 (with 256 pair slots: 40 KB of LDS per workgroup, 4 workgroups per CU; block M)
#endif
#ifndef DT_TRI_PAD
#define DT_TRI_PAD 0.5f        // round 3: was 1 px; a quarter fewer box candidates per pixel in k_resolve_obj's z-buffer, same frames
#endif
#define TRI_CAP DT_TRI_CAP  // LDS triangle slots per wavefront in k_resolve_obj (streamed chunks)

#define CLS_SKY 0
#define CLS_GROUND 1
#define CLS_TILE 2

#define NEAR_Z 0.04f
#define FAR_Z 100.0f
#define GROUND_Y (-0.008f)   // ground quad: y=-0.8 scaled by 0.01 (simulator.py:510-526,1810)
#define GROUND_HALF 50.0f

struct alignas(16) EnvCam {          // 32 floats = 128 B, written by k_cam_setup
  float Cx, Cy, Cz;      // camera centre (world)
  float sa, ca;          // yaw
  float sth, cth;        // pitch (cam_angle[0])
  float tx, ty;          // tan(fov_y/2)*aspect, tan(fov_y/2)
  float hor[3];          // horizon colour * 255
  float gnd[3];          // ground colour * 255
  float base[3];         // scene ambient 0.3 + light ambient
  float dif[3];          // light diffuse
  float L[4];            // light position, eye space (w=0: direction, pre-normalised)
  float gndl[4];         // max(0, N.L) at the 4 ground-quad corners (-x-z, +x-z, -x+z, +x+z)
  int32_t map_id;
  float pad[2];
};
static_assert(sizeof(EnvCam) == 128, "EnvCam is 128 bytes");

// Everything the fast path needs per env, fully precomputed by k_cam_setup so that the env loop of
// k_raster does no wave-uniform arithmetic on the vector ALU (there is no scalar float ALU on gfx950):
// one 64-byte scalar load per env.
struct alignas(16) EnvFast {
  float A, B, Cxi, Czi;        // yaw rotation straight into tile units: gx = Cxi + lr*A + lf*B, gz = Czi + lr*B - lf*A
  float kg, Cx, Cz, its;       // ground-plane scale (Cy+0.008)/Cy, camera centre, 1/tile_size
  float ts, gw_m, gh_m, I0;    // tile size, grid extent in metres, shared-camera light base (unused with DR)
  uint32_t hor_rgb;            // packed horizon colour
  int32_t gw, gh, tile_off;    // grid size, first tile record of the env's map
};
static_assert(sizeof(EnvFast) == 64, "EnvFast is 64 bytes");

// Per-env constants of the quad-layout fast path (k_raster_q): the tile-plane hit (lr, lf) of a pixel in the yaw-local
// frame goes straight to padded quad coordinates  X = Cx + lr*A + lf*B,  Z = Cz + lr*B - lf*A  (tile units x S, +0.5
// for the GL_LINEAR half-texel shift, + DT_QRING tiles of off-grid ring).  One 64-byte scalar load per env.
struct alignas(16) EnvQ {
  float A, B, Cx, Cz;
  float Xhi, Zhi;              // clamp range [S/2, hi]: centres of the outermost ring cells
  uint32_t tab_b, pitch4;      // the map's padded tile table inside the LDS copy (byte offset), row pitch in bytes (8-byte entries)
  uint32_t hor_rgb;            // packed horizon colour
  uint32_t sky[3];             // the horizon colour as the three dwords of four consecutive pixels (RGBR GBRG BRGB)
  uint32_t reach;              // cells from the camera to the border of the padded grid: hits closer than that need no clamp
  uint32_t env;                // the env (frame index) at this position of the render order (k_env_sort)
  uint32_t pad[2];             // [0]: k_raster_v3's LDS column offset of the map, [1]: quad cells per metre (float bits)
};
static_assert(sizeof(EnvQ) == 64, "EnvQ is 64 bytes");
// k_raster_v3's per-env constants, in render order: the transform as PAIRS (the scalar operand of a v_pk_fma_f32 is two
// consecutive scalar registers: loaded as pairs they need no s_mov), everything its env loop reads in one 64-byte scalar
// load.  [N + 1] entries: the loop prefetches one past its chunk.
struct alignas(16) EnvV {
  float A2[2], B2[2], Cx2[2], Cz2[2];
  float Xhi, Zhi;
  uint32_t tab3, hor_rgb, reach, env, pad[2];
};
static_assert(sizeof(EnvV) == 64, "EnvV is 64 bytes");

// coverage-only part of a ScreenTri kept in LDS by k_resolve_obj; the winner's colours are fetched
// from global memory.
struct alignas(16) TriCov { float bx0, bx1, by0, by1; float sx[3], sy[3], iw[3]; float inv_area; int32_t index; int32_t pad; };   // == first half of ScreenTri
static_assert(sizeof(TriCov) == 64, "TriCov is 64 bytes");

namespace {

// Camera intrinsics / light shared by all envs when DTSIM_F_DOMAIN_RAND is off
// (simulator.py:119-131 CAMERA_ANGLE / CAMERA_FOV_Y / CAMERA_FLOOR_DIST, :570-576).
struct CamShared { float sth, cth, tx, ty, Cy, base, dif; float L[4]; };

__device__ inline CamShared default_cam(float aspect) {
  CamShared s;
  const float th = 19.15f * 0.017453292519943295f;
  s.sth = sinf(th); s.cth = cosf(th);
  s.ty = tanf(0.5f * 75.0f * 0.017453292519943295f); s.tx = s.ty * aspect;
  s.Cy = 0.108f; s.base = 0.3f + 0.25f; s.dif = 0.35f;
  s.L[0] = 0.f; s.L[1] = 3.f; s.L[2] = 0.f; s.L[3] = 1.f;
  return s;
}

struct EnvD;                                          // render_v3dr.inc: per-env constants of k_raster_v3dr
__device__ inline void fill_envd_at(EnvD* arr, int idx, const EnvCam& c, const EnvQ& q, const RenderMapDev& m, float q_cells, int H, int W);

// Render order of the envs for the quad-layout path: envs standing on the same tile (and facing the same way) are
// made neighbours, so that the 64 envs a raster workgroup loops over -- and, with the XCD-affine workgroup map of
// k_raster_q, all the envs one XCD's L2 serves -- look at the same few texture blocks.  A counting sort by
// (tile under the camera, heading quadrant) in one workgroup; the order inside a bin is arbitrary (frames are
// independent, so the order never changes a result).  pos[e] = position of env e.
#define SORT_BINS 4096
__global__ __launch_bounds__(1024) void k_env_sort(SimArrays A, const RenderMapDev* __restrict__ maps, int32_t* __restrict__ pos) {
  __shared__ int s_hist[SORT_BINS];
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < SORT_BINS; i += 1024) s_hist[i] = 0;
  __syncthreads();
  // (any order is a valid render order: the bin only needs to be the same in both passes -- single-precision trigonometry,
  // computed once per env and kept in registers between the histogram and the scatter; N <= 8 * 1024 envs per launch keep
  // theirs, larger batches recompute)
  auto bin_of = [&](int e) -> int {
    const int mid = A.map_id[e] < 0 ? 0 : A.map_id[e];
    const float its = maps[mid].inv_tile_size;
    const float ang = (float)A.angle[e];
    float sa, ca;
    __sincosf(ang, &sa, &ca);
    const float px = (float)A.pos_x[e] + (float)DT_CAMERA_FORWARD_DIST * ca, pz = (float)A.pos_z[e] - (float)DT_CAMERA_FORWARD_DIST * sa;
    const int ti = min(max((int)floorf(px * its), 0), 31), tj = min(max((int)floorf(pz * its), 0), 31);
    const int quad = ((int)floorf(ang * (float)(2.0 / 3.141592653589793) + 0.5f)) & 3;
    return ((((tj << 5) | ti) << 2) | quad) ^ ((mid * 1237) & (SORT_BINS - 1));
  };
  constexpr int KEEP = 8;
  int bins[KEEP];
#pragma unroll
  for (int k = 0; k < KEEP; ++k) {
    const int e = tid + k * 1024;
    bins[k] = e < A.N ? bin_of(e) : -1;
    if (bins[k] >= 0) atomicAdd(&s_hist[bins[k]], 1);
  }
  for (int e = tid + KEEP * 1024; e < A.N; e += 1024) atomicAdd(&s_hist[bin_of(e)], 1);
  __syncthreads();
  // exclusive scan of the bins: each thread owns SORT_BINS / 1024 consecutive bins
  int loc[SORT_BINS / 1024], sum = 0;
#pragma unroll
  for (int k = 0; k < SORT_BINS / 1024; ++k) { loc[k] = sum; sum += s_hist[tid * (SORT_BINS / 1024) + k]; }
  // block-wide exclusive scan of the 1024 partial sums: within a wavefront by DPP-style shuffles, across the 16 wavefronts through LDS
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if ((tid & 63) >= d) incl += v; }
  if ((tid & 63) == 63) s_part[tid >> 6] = incl;
  __syncthreads();
  if (tid < 16) {
    int w = s_part[tid];
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(w, d, 16); if (tid >= d) w += v; }
    s_part[16 + tid] = w;                             // inclusive over the wavefronts
  }
  __syncthreads();
  const int base = incl - sum + ((tid >> 6) ? s_part[16 + (tid >> 6) - 1] : 0);
#pragma unroll
  for (int k = 0; k < SORT_BINS / 1024; ++k) s_hist[tid * (SORT_BINS / 1024) + k] = base + loc[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KEEP; ++k)
    if (bins[k] >= 0) pos[tid + k * 1024] = atomicAdd(&s_hist[bins[k]], 1);
  for (int e = tid + KEEP * 1024; e < A.N; e += 1024) pos[e] = atomicAdd(&s_hist[bin_of(e)], 1);
}

__global__ void k_cam_setup(SimArrays A, int domain_rand, int segment, float aspect, EnvCam* out, EnvFast* fast,
                            const RenderMapDev* __restrict__ maps, EnvQ* envq, int qlog2, const int32_t* __restrict__ pos, EnvV* envv,
                            EnvD* envd, int W, int H) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t N = A.N;
  if (e >= A.N) return;
  EnvCam c;
  const double ang = A.angle[e];
  const double sa = sin(ang), ca = cos(ang);
  double px = A.pos_x[e], py = 0.0, pz = A.pos_z[e];
  float base[3], dif[3], L[4];
  if (domain_rand) {  // per-env camera / light (simulator.py:565-614, 1768-1769)
    px += (double)A.cam[3 * N + e]; py += (double)A.cam[4 * N + e]; pz += (double)A.cam[5 * N + e];
    py += (double)A.cam[0 * N + e];
    const float th = A.cam[1 * N + e], fov = A.cam[2 * N + e];
    for (int k = 0; k < 3; ++k) { base[k] = 0.3f + A.colors[(6 + k) * N + e]; dif[k] = A.colors[(9 + k) * N + e]; }
    for (int k = 0; k < 4; ++k) L[k] = A.colors[(12 + k) * N + e];
    c.sth = sinf(th); c.cth = cosf(th);
    const float tanh_ = tanf(0.5f * fov);
    c.tx = tanh_ * aspect; c.ty = tanh_;
    c.Cy = (float)py;
  } else {
    const CamShared s = default_cam(aspect);
    c.sth = s.sth; c.cth = s.cth; c.tx = s.tx; c.ty = s.ty; c.Cy = s.Cy;
    for (int k = 0; k < 3; ++k) { base[k] = s.base; dif[k] = s.dif; }
    for (int k = 0; k < 4; ++k) L[k] = s.L[k];
  }
  // glTranslatef(0,0,CAMERA_FORWARD_DIST) before gluLookAt (simulator.py:1784,1803): the
  // camera centre sits 6.6 cm ahead of the axle along dir = (cos a, 0, -sin a).
  c.Cx = (float)(px + DT_CAMERA_FORWARD_DIST * ca);
  c.Cz = (float)(pz - DT_CAMERA_FORWARD_DIST * sa);
  c.sa = (float)sa; c.ca = (float)ca;
  for (int k = 0; k < 3; ++k) {
    c.hor[k] = 255.f * A.colors[(0 + k) * N + e];
    c.gnd[k] = 255.f * A.colors[(3 + k) * N + e];
    c.base[k] = base[k];     // GL_LIGHT_MODEL_AMBIENT 0.3 (simulator.py:1741) + light ambient
    c.dif[k] = dif[k];
  }
  if (segment) {             // simulator.py:1730-1733 lighting off; :1753 clear and :1808 ground quad magenta
    for (int k = 0; k < 3; ++k) { c.base[k] = 1.f; c.dif[k] = 0.f; c.hor[k] = c.gnd[k] = (k == 1) ? 0.f : 255.f; }
  }
  if (L[3] == 0.f) {  // directional: normalise once
    const float inv = rsqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
    L[0] *= inv; L[1] *= inv; L[2] *= inv;
  }
  for (int k = 0; k < 4; ++k) c.L[k] = L[k];
  // ground quad corners: per-vertex lighting (Gouraud over the 100 m quad)
  for (int k = 0; k < 4; ++k) {
    const float X = (k & 1) ? GROUND_HALF : -GROUND_HALF, Z = (k & 2) ? GROUND_HALF : -GROUND_HALF;
    const float rx = X - c.Cx, ry = GROUND_Y - c.Cy, rz = Z - c.Cz;
    const float xla = rx * c.sa + rz * c.ca;          // . right = (sin a, 0, cos a)
    const float zla = -(rx * c.ca - rz * c.sa);       // -(. dir)
    const float ye = ry * c.cth - zla * c.sth, ze = ry * c.sth + zla * c.cth, xe = xla;
    // The ground vertex list has no normals (simulator.py:526): GL lights it with the CURRENT normal -- (0, 0, 1), GL's initial value;
    // nothing on the reference's default path calls glNormal -- taken through the inverse transpose of glScalef(50, 0.01, 50) (:1810)
    // and the view rotation, NOT renormalised (GL_NORMALIZE is off): length 1/50.  Pinned on Mesa llvmpipe (tests/test_gpu_gl_golden.py).
    const float gnx = c.ca * (1.f / GROUND_HALF), gny = -c.sa * c.sth * (1.f / GROUND_HALF), gnz = c.sa * c.cth * (1.f / GROUND_HALF);
    float ndl;
    if (L[3] == 0.f) ndl = gnx * L[0] + gny * L[1] + gnz * L[2];
    else {
      const float lx = L[0] - xe, ly = L[1] - ye, lz = L[2] - ze;
      ndl = (gnx * lx + gny * ly + gnz * lz) * rsqrtf(lx * lx + ly * ly + lz * lz);
    }
    c.gndl[k] = fmaxf(ndl, 0.f);
  }
  c.map_id = A.map_id[e];
  c.pad[0] = c.pad[1] = 0.f;
  out[e] = c;
  const RenderMapDev m = maps[c.map_id];
  EnvFast f;
  f.its = m.inv_tile_size; f.ts = m.tile_size;
  f.A = c.sa * f.its; f.B = c.ca * f.its; f.Cxi = c.Cx * f.its; f.Czi = c.Cz * f.its;
  f.kg = (c.Cy - GROUND_Y) / c.Cy; f.Cx = c.Cx; f.Cz = c.Cz;
  f.gw_m = (float)m.grid_w * m.tile_size; f.gh_m = (float)m.grid_h * m.tile_size;
  f.I0 = 0.f;
  {
    const uint32_t r = (uint32_t)(fminf(fmaxf(c.hor[0], 0.f), 255.f) + 0.5f), g = (uint32_t)(fminf(fmaxf(c.hor[1], 0.f), 255.f) + 0.5f);
    const uint32_t b = (uint32_t)(fminf(fmaxf(c.hor[2], 0.f), 255.f) + 0.5f);
    f.hor_rgb = r | (g << 8) | (b << 16);
  }
  f.gw = m.grid_w; f.gh = m.grid_h; f.tile_off = m.tile_off;
  fast[e] = f;
  if (envq) {
    const float S = (float)(1 << qlog2);
    EnvQ q;
    q.A = f.A * S; q.B = f.B * S;
    q.Cx = f.Cxi * S + 0.5f + (float)DT_QRING * S; q.Cz = f.Czi * S + 0.5f + (float)DT_QRING * S;
    q.Xhi = ((float)(m.grid_w + 2 * DT_QRING) - 0.5f) * S; q.Zhi = ((float)(m.grid_h + 2 * DT_QRING) - 0.5f) * S;
    q.tab_b = (uint32_t)m.qt_off * 8u; q.pitch4 = (uint32_t)m.qt_pitch * 8u;   // 8-byte table entries
    q.hor_rgb = f.hor_rgb;
    const uint32_t r = f.hor_rgb & 255u, g = (f.hor_rgb >> 8) & 255u, b = (f.hor_rgb >> 16) & 255u;
    q.sky[0] = r | (g << 8) | (b << 16) | (r << 24);
    q.sky[1] = g | (b << 8) | (r << 16) | (g << 24);
    q.sky[2] = b | (r << 8) | (g << 16) | (b << 24);
    {
      const float xmax = (float)(m.grid_w + 2 * DT_QRING) * S, zmax = (float)(m.grid_h + 2 * DT_QRING) * S;
      const float r = fminf(fminf(q.Cx, xmax - q.Cx), fminf(q.Cz, zmax - q.Cz)) - 2.f;
      q.reach = r > 0.f ? (uint32_t)fminf(r, 1.0e9f) : 0u;      // NaN / outside the padded grid -> 0: always clamp
    }
    q.env = (uint32_t)e; q.pad[1] = __float_as_uint(m.inv_tile_size * S);   // quad cells per metre of the env's map (the exact path's ground-quad test)
    q.pad[0] = (uint32_t)c.map_id * (32u * 4u);     // k_raster_v3: byte offset of the map's columns inside an LDS table row (V3_MAP_COLS entries)
    envq[pos ? pos[e] : e] = q;
    if (envv) {
      EnvV v;
      v.A2[0] = v.A2[1] = q.A; v.B2[0] = v.B2[1] = q.B; v.Cx2[0] = v.Cx2[1] = q.Cx; v.Cz2[0] = v.Cz2[1] = q.Cz;
      v.Xhi = q.Xhi; v.Zhi = q.Zhi; v.tab3 = q.pad[0]; v.hor_rgb = q.hor_rgb; v.reach = q.reach; v.env = q.env; v.pad[0] = v.pad[1] = 0u;
      envv[pos ? pos[e] : e] = v;
      if (e == A.N - 1) envv[A.N] = v;               // the entry past the end (prefetched, never used)
    }
    if (envd) fill_envd_at(envd, pos ? pos[e] : e, c, q, m, S, H, W);   // domain randomisation on the quad records (index = position in the render order)
  }
}

// ---- mesh objects -> per-env screen-space triangles ------------------------------------
// One workgroup per env.  WorldObj.render (objects.py:123-148): T(pos) S(scale) Ry(y_rot), mesh
// chunks with per-vertex Kd colour (objmesh.py:241-293), GL per-vertex lighting (normals through the inverse transpose of the
// model-view, NOT renormalised: divided by the scale -- DESIGN.md "Render spec"), projected to rectilinear pixel coordinates.
// order-preserving float <-> int map (for integer atomics on floats)
__device__ inline int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ inline float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Source-pixel bounding box of every raster wavefront block (tile * 4 + wavefront): env-invariant, one workgroup per
// raster tile.  Empty blocks (no pixel inside the rectilinear image) get an inverted box.
__global__ __launch_bounds__(RB) void k_blk_setup(RenderParams R, const float4* __restrict__ lut, float4* __restrict__ blockbox) {
  const int tiles_x = (R.W + DT_TILE_W - 1) / DT_TILE_W;
  const int tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile_x0 = (tile % tiles_x) * DT_TILE_W;
  const int wave_y0 = (tile / tiles_x) * DT_TILE_H + wave * (WAVE_PIX / WAVE_W);
  float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int x = tile_x0 + SLOT_X(k, lane), y = wave_y0 + SLOT_Y(k, lane);
    if (x < R.W && y < R.H) {
      const float4 l = lut[y * R.W + x];
      if (l.z != 0.f) {
        const float sx = (l.x + 1.f) * 0.5f * (float)R.W, sy = (1.f - l.y) * 0.5f * (float)R.H;
        x0 = fminf(x0, sx); x1 = fmaxf(x1, sx); y0 = fminf(y0, sy); y1 = fmaxf(y1, sy);
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, d)); x1 = fmaxf(x1, __shfl_xor(x1, d));
    y0 = fminf(y0, __shfl_xor(y0, d)); y1 = fmaxf(y1, __shfl_xor(y1, d));
  }
  if (lane == 0) blockbox[tile * 4 + wave] = make_float4(x0, x1, y0, y1);
  if (blockIdx.x == 0) {   // triangle range of every object in its map's triangle order (static; k_resolve_obj streams by it)
    for (int i = threadIdx.x; i < R.n_maps * DTSIM_MAX_OBJECTS; i += RB) {
      const int mi = i / DTSIM_MAX_OBJECTS, o = i % DTSIM_MAX_OBJECTS;
      const RenderMapDev m = R.maps[mi];
      uint2 fc = make_uint2(0u, 0u);
      if (o < m.n_obj) {
        for (int k = 0; k < o; ++k) { const int mid = R.objs[m.obj_off + k].mesh_id; fc.x += mid >= 0 ? (uint32_t)R.meshes[mid].n_tris : 0u; }
        const int mid = R.objs[m.obj_off + o].mesh_id;
        fc.y = mid >= 0 ? (uint32_t)R.meshes[mid].n_tris : 0u;
      }
      R.objrange[i] = fc;
    }
  }
}

// ---- per-map constants the raster needs (wave-uniform) -------------------------------
struct MapU { float its, ts, gwf, ghf; int gw, gh, tile_off; };

__device__ inline MapU map_u(const RenderMapDev& m) {
  MapU u;
  u.its = m.inv_tile_size; u.ts = m.tile_size; u.gw = m.grid_w; u.gh = m.grid_h;
  u.gwf = (float)m.grid_w; u.ghf = (float)m.grid_h; u.tile_off = m.tile_off;
  return u;
}

// 1-ulp hardware reciprocal / rsqrt / sqrt: a relative error of 1e-7 moves a ground hit by < 1e-3 texel
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline float fsqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }

struct Ray { float xe, ye, yla, fwd; };

__device__ inline Ray make_ray(float nx, float ny, float tx, float ty, float sth, float cth) {
  Ray r;
  r.xe = nx * tx; r.ye = ny * ty;
  r.yla = r.ye * cth - sth;      // world-up component of the ray
  r.fwd = r.ye * sth + cth;      // component along dir
  return r;
}

// positional / directional light on a surface with eye-space normal (0, cth, sth) at the
// eye-space point t*(xe, ye, -1): max(0, N.L)   (tiles; simulator.py:565-591)
__device__ inline float plane_ndl(const float L[4], float sth, float cth, const Ray& r, float t) {
  float ndl;
  if (L[3] == 0.f) ndl = cth * L[1] + sth * L[2];
  else {
    const float lx = L[0] - t * r.xe, ly = L[1] - t * r.ye, lz = L[2] + t;
    ndl = (cth * ly + sth * lz) * frsq(lx * lx + ly * ly + lz * lz);
  }
  return fmaxf(ndl, 0.f);
}

// ---- GL_LINEAR as the reference's renderer computes it -------------------------------------------------------------------
// GL leaves the filter's precision to the implementation.  The reference's CI renderer -- Mesa llvmpipe, the one the GL goldens
// come from (tests/golden/ref_gl_*.npz) -- filters RGBA8 textures in integers (measured bit-exact, profiles/r06_gl_filter_precision.txt):
// texel coordinate * 256, rounded, minus half a texel; the low 8 bits are the weight; lerp(w, p, q) = p + ((w (q - p) + 128) >> 8),
// along s for both rows, then along t on the 8-bit results.  gl_fix: coordinate already shifted by the half texel -> fixed point.
__device__ inline int gl_lerp8(int w, int p, int q) { return p + ((w * (q - p) + 128) >> 8); }
__device__ inline int gl_fix(float x) { return (int)floorf(fmaf(x, 256.f, 0.5f)); }
__device__ inline void gl_linear_rgb(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int wx, int wy, int out[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c00 = (int)((t00 >> (8 * k)) & 255u), c10 = (int)((t10 >> (8 * k)) & 255u);
    const int c01 = (int)((t01 >> (8 * k)) & 255u), c11 = (int)((t11 >> (8 * k)) & 255u);
    out[k] = gl_lerp8(wy, gl_lerp8(wx, c00, c10), gl_lerp8(wx, c01, c11));
  }
}

// GL_LINEAR/GL_REPEAT fetch from the padded texture (the exact paths and the generic raster: llvmpipe's arithmetic), times the lit
// vertex colour I.
__device__ inline void tile_color(const RenderParams& R, const TileLds& tr, float fx, float fz, const float I[3],
                                  float out[3]) {
  if (!(tr.flags & 2u)) {  // untextured tile: white vertex colour
    out[0] = 255.f * I[0]; out[1] = 255.f * I[1]; out[2] = 255.f * I[2];
    return;
  }
  const float x = fmaf(tr.mxz, fz, fmaf(tr.mxx, fx, tr.ox)), y = fmaf(tr.myz, fz, fmaf(tr.myx, fx, tr.oy));
  const int xf = gl_fix(x), yf = gl_fix(y);
  const int x0 = (xf >> 8) & (R.tex_w - 1), y0 = (yf >> 8) & (R.tex_h - 1);
  const uint32_t* pt = R.texels + tr.tex_off + y0 * (R.tex_w + 1) + x0;
  uint2 top2, bot2;               // two 8-byte loads: (x0,y0),(x0+1,y0) and the row above
  __builtin_memcpy(&top2, pt, 8);
  __builtin_memcpy(&bot2, pt + (R.tex_w + 1), 8);
  int T[3];
  gl_linear_rgb(top2.x, top2.y, bot2.x, bot2.y, xf & 255, yf & 255, T);
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = (float)T[k] * I[k];
}

// GL_LINEAR / GL_REPEAT fetch of a mesh texture at (u, v) in texture coordinates; out in 0..1 (the sampler's 8-bit result / 255).
// Any power-of-two size; storage is the padded (h+1) x (w+1) layout of the texel pool.
__device__ inline void mesh_texel(const RenderParams& R, int tex, float u, float v, float out[3]) {
  const TexDev td = R.tex[tex];
  const int xf = gl_fix(u * (float)td.w) - 128, yf = gl_fix(v * (float)td.h) - 128;
  const int x0 = (xf >> 8) & (td.w - 1), y0 = (yf >> 8) & (td.h - 1);
  const uint32_t* pt = R.texels + td.off + y0 * (td.w + 1) + x0;
  int T[3];
  gl_linear_rgb(pt[0], pt[1], pt[td.w + 1], pt[td.w + 2], xf & 255, yf & 255, T);
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = (float)T[k] * (1.f / 255.f);
}

__device__ inline float ground_ndl(const EnvCam& c, float wx, float wz) {
  const float a = fminf(fmaxf((wx + GROUND_HALF) * (0.5f / GROUND_HALF), 0.f), 1.f);
  const float b = fminf(fmaxf((wz + GROUND_HALF) * (0.5f / GROUND_HALF), 0.f), 1.f);
  const float n0 = c.gndl[0] + a * (c.gndl[1] - c.gndl[0]);
  const float n1 = c.gndl[2] + a * (c.gndl[3] - c.gndl[2]);
  return n0 + b * (n1 - n0);
}


#ifndef DT_OBJSETUP_T
#define DT_OBJSETUP_T 256           // threads of a k_obj_setup workgroup (one workgroup per env)
#endif
__global__ __launch_bounds__(DT_OBJSETUP_T) void k_obj_setup(SimArrays A, RenderParams R, const EnvCam* __restrict__ cams, const int32_t* __restrict__ pos) {
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t N = A.N;
  const EnvCam c = cams[e];
  const RenderMapDev m = R.maps[c.map_id];
  ScreenTri* out = R.stris + (size_t)e * R.max_tris;
  // per-object screen boxes: LDS min / max through the order-preserving float -> int map
  __shared__ int s_obox[DTSIM_MAX_OBJECTS][4];
  __shared__ float s_oboxf[DTSIM_MAX_OBJECTS][4];
  if (tid < DTSIM_MAX_OBJECTS) {
    s_obox[tid][0] = s_obox[tid][2] = 0x7fffffff;    // min slots
    s_obox[tid][1] = s_obox[tid][3] = (int)0x80000000;   // max slots
  }
  // Round 5: object-level frustum cull first.  An env sees ~ 1 of its map's objects (profiles/r04_variants_ab.txt block G: 0.8 live objects per
  // env); an object whose model-space box, taken through the instance and the camera, lies wholly outside one frustum plane (or is invisible)
  // contributes no triangle: its triangles are neither loaded nor transformed nor written -- nothing reads them, the object's screen box
  // stays empty and so it appears in no block mask (k_resolve_obj streams the triangles of masked objects only).
  __shared__ uint8_t s_objlive[DTSIM_MAX_OBJECTS];
  __shared__ int s_nt[DTSIM_MAX_OBJECTS], s_first[DTSIM_MAX_OBJECTS + 1];   // triangles of each object, and before it (the walks below read LDS, not dependent global loads)
  if (tid < DTSIM_MAX_OBJECTS) {
    bool livef = false;
    s_nt[tid] = 0;
    if (tid < m.n_obj) {
      const ObjInstDev oi = R.objs[m.obj_off + tid];
      s_nt[tid] = oi.mesh_id >= 0 ? R.meshes[oi.mesh_id].n_tris : 0;
      if (oi.mesh_id >= 0 && A.ob_visible[(size_t)tid * N + e] != 0) {
        const MeshDev md = R.meshes[oi.mesh_id];
        float px = oi.x, py = oi.y, pz = oi.z, yrot = oi.yrot_deg;
        if (oi.dyn_slot >= 0) {
          px = (float)A.ob_cx[(size_t)oi.dyn_slot * N + e]; pz = (float)A.ob_cz[(size_t)oi.dyn_slot * N + e];
          py += (float)A.ob_cy[(size_t)oi.dyn_slot * N + e];
          yrot = (float)A.ob_yrot[(size_t)oi.dyn_slot * N + e];
        }
        const float ang = yrot * 0.017453292519943295f;
        const float co = cosf(ang), so = sinf(ang);
        uint32_t all_out = 0x1Fu;                       // bit set while every corner so far is outside that plane: near, left, right, bottom, top
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float mx = ((k & 1) ? md.mx[0] : md.mn[0]) * oi.scale, my = ((k & 2) ? md.mx[1] : md.mn[1]) * oi.scale, mz = ((k & 4) ? md.mx[2] : md.mn[2]) * oi.scale;
          const float rx = (mx * co + mz * so + px) - c.Cx, ry = (my + py) - c.Cy, rz = (-mx * so + mz * co + pz) - c.Cz;
          const float xla = rx * c.sa + rz * c.ca, zla = -(rx * c.ca - rz * c.sa);
          const float xe = xla, ye = ry * c.cth - zla * c.sth, w = -(ry * c.sth + zla * c.cth);
          // the four side planes pass through the eye: xe + w tx >= 0 ... are half-spaces for any sign of w; two pixels of slack (the
          // triangle boxes are padded by DT_TRI_PAD)
          const float lx = w * c.tx, ly = w * c.ty, sx_ = fabsf(w) * c.tx * (4.f / (float)R.W), sy_ = fabsf(w) * c.ty * (4.f / (float)R.H);
          uint32_t o = 0u;
          if (w <= NEAR_Z) o |= 1u;
          if (xe < -lx - sx_) o |= 2u;
          if (xe > lx + sx_) o |= 4u;
          if (ye < -ly - sy_) o |= 8u;
          if (ye > ly + sy_) o |= 16u;
          all_out &= o;
        }
        livef = all_out == 0u;
      }
    }
    s_objlive[tid] = livef ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int o = 0; o < m.n_obj; ++o) { s_first[o] = acc; acc += s_nt[o]; }
    s_first[m.n_obj] = acc;
  }
  __syncthreads();
  int obj = 0;                             // walk the object list as t grows (t is monotone per thread)
  for (int t = tid; t < m.n_tris; t += DT_OBJSETUP_T) {
    while (obj + 1 < m.n_obj && t >= s_first[obj + 1]) ++obj;
    if (!s_objlive[obj]) continue;        // the whole object is out of view (or invisible)
    const int obj_first = s_first[obj];
    const ObjInstDev oi = R.objs[m.obj_off + obj];
    const TriDev td = R.tris[R.meshes[oi.mesh_id].off + (t - obj_first)];
    float px = oi.x, py = oi.y, pz = oi.z, yrot = oi.yrot_deg;
    if (oi.dyn_slot >= 0) {               // DuckieObj: pos = center, y_rot wiggles (objects.py:408-410)
      px = (float)A.ob_cx[(size_t)oi.dyn_slot * N + e]; pz = (float)A.ob_cz[(size_t)oi.dyn_slot * N + e];
      py += (float)A.ob_cy[(size_t)oi.dyn_slot * N + e];
      yrot = (float)A.ob_yrot[(size_t)oi.dyn_slot * N + e];
    }
    const bool visible = A.ob_visible[(size_t)obj * N + e] != 0;
    const float ang = yrot * 0.017453292519943295f;
    const float co = cosf(ang), so = sinf(ang);
    ScreenTri st;
    bool ok = visible;
    float w[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const float mx = td.v[v][0] * oi.scale, my = td.v[v][1] * oi.scale, mz = td.v[v][2] * oi.scale;
      const float rx = (mx * co + mz * so + px) - c.Cx, ry = (my + py) - c.Cy, rz = (-mx * so + mz * co + pz) - c.Cz;
      const float xla = rx * c.sa + rz * c.ca, zla = -(rx * c.ca - rz * c.sa);
      const float xe = xla, ye = ry * c.cth - zla * c.sth, ze = ry * c.sth + zla * c.cth;
      w[v] = -ze;
      ok = ok && (w[v] > NEAR_Z);
      // normal: Ry(y_rot) then view rotation, through the inverse transpose of glScalef(scale) (objects.py:142) = divided by the scale, and NOT
      // renormalised (GL_NORMALIZE / GL_RESCALE_NORMAL are never enabled): whatever length the OBJ file gives it stays.  Pinned on Mesa llvmpipe.
      const float nwx = td.n[v][0] * co + td.n[v][2] * so, nwy = td.n[v][1], nwz = -td.n[v][0] * so + td.n[v][2] * co;
      const float nxl = nwx * c.sa + nwz * c.ca, nzl = -(nwx * c.ca - nwz * c.sa);
      float nex = nxl, ney = nwy * c.cth - nzl * c.sth, nez = nwy * c.sth + nzl * c.cth;
      const float ninv = 1.f / oi.scale;
      nex *= ninv; ney *= ninv; nez *= ninv;
      float ndl;
      if (c.L[3] == 0.f) ndl = nex * c.L[0] + ney * c.L[1] + nez * c.L[2];
      else {
        const float lx = c.L[0] - xe, ly = c.L[1] - ye, lz = c.L[2] - ze;
        ndl = (nex * lx + ney * ly + nez * lz) * rsqrtf(lx * lx + ly * ly + lz * lz);
      }
      ndl = fmaxf(ndl, 0.f);
      const float iw = 1.f / w[v];
      st.iw[v] = iw;
      st.sx[v] = (xe * iw / c.tx + 1.f) * 0.5f * (float)R.W;
      st.sy[v] = (1.f - ye * iw / c.ty) * 0.5f * (float)R.H;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        st.cw[v][k] = fminf(td.c[v][k] * (c.base[k] + c.dif[k] * ndl), 1.f) * 255.f * iw;
      st.uw[v] = td.uv[v][0] * iw; st.vw[v] = td.uv[v][1] * iw;
    }
    st.tex = td.tex; st.pad = 0;
    if (oi.light_tris > 0 && (t - obj_first) < oi.light_tris)          // traffic light card: texture by pattern
      st.tex = A.ob_light[(size_t)obj * N + e] ? oi.light_tex1 : oi.light_tex0;
    if (R.segment) {         // get_mesh(name, segment=True): one flat colour as every chunk's texture (objmesh.py:255-292),
      const uint8_t* sc = R.mesh_seg + 4 * oi.mesh_id;   // MODULATEd with the per-vertex Kd -> fold it into the vertex colours
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int k = 0; k < 3; ++k) st.cw[v][k] *= (float)sc[k] * (1.f / 255.f);
      st.tex = -1;
    }
    const float area = (st.sx[1] - st.sx[0]) * (st.sy[2] - st.sy[0]) - (st.sx[2] - st.sx[0]) * (st.sy[1] - st.sy[0]);
    ok = ok && (area != 0.f);
    st.inv_area = ok ? 1.f / area : 0.f;
    // screen box padded by DT_TRI_PAD pixels: the four samples sit within 0.375 px of the pixel centre the box is tested with
    st.bx0 = fminf(fminf(st.sx[0], st.sx[1]), st.sx[2]) - DT_TRI_PAD; st.bx1 = fmaxf(fmaxf(st.sx[0], st.sx[1]), st.sx[2]) + DT_TRI_PAD;
    st.by0 = fminf(fminf(st.sy[0], st.sy[1]), st.sy[2]) - DT_TRI_PAD; st.by1 = fmaxf(fmaxf(st.sy[0], st.sy[1]), st.sy[2]) + DT_TRI_PAD;
    if (ok && (st.bx1 < 0.f || st.bx0 > (float)R.W || st.by1 < 0.f || st.by0 > (float)R.H)) { ok = false; st.inv_area = 0.f; }
    if (!ok) { st.bx0 = 1e30f; st.bx1 = -1e30f; st.by0 = 1e30f; st.by1 = -1e30f; }
    st.index = t;
    // culled triangles (outside the view, degenerate, beyond near / far) leave only their inverted box: k_resolve_obj culls on
    // the boxes and reads the 128-byte record of survivors only
    if (ok || !R.tribox) out[t] = st;
    if (R.tribox) R.tribox[(size_t)e * R.max_tris + t] = make_float4(st.bx0, st.bx1, st.by0, st.by1);   // the cull stream of k_resolve_obj: 16 B per triangle, contiguous
    if (ok) {
      atomicMin(&s_obox[obj][0], f2ord(st.bx0)); atomicMax(&s_obox[obj][1], f2ord(st.bx1));
      atomicMin(&s_obox[obj][2], f2ord(st.by0)); atomicMax(&s_obox[obj][3], f2ord(st.by1));
    }
  }
  __syncthreads();
  if (tid < m.n_obj) {                      // per-object screen box = union of its live triangles' boxes
    ObjBox ob;
    ob.first = s_first[tid]; ob.pad[0] = ob.pad[1] = 0;
    const int cnt = s_nt[tid];
    const bool live = s_obox[tid][0] != 0x7fffffff;
    ob.bx0 = live ? ord2f(s_obox[tid][0]) : 1e30f; ob.bx1 = live ? ord2f(s_obox[tid][1]) : -1e30f;
    ob.by0 = live ? ord2f(s_obox[tid][2]) : 1e30f; ob.by1 = live ? ord2f(s_obox[tid][3]) : -1e30f;
    ob.count = live ? cnt : 0;
    R.objbox[(size_t)e * DTSIM_MAX_OBJECTS + tid] = ob;
    s_oboxf[tid][0] = ob.bx0; s_oboxf[tid][1] = ob.bx1; s_oboxf[tid][2] = ob.by0; s_oboxf[tid][3] = ob.by1;   // empty when not live
  }
  __syncthreads();
  if (R.objmask) {
    // which objects' screen boxes meet each raster wavefront block (source-pixel boxes of the blocks: k_blk_setup): the
    // raster reads one 8-byte mask per (env, block) instead of walking the env's object boxes
    const int n_blk = ((R.W + DT_TILE_W - 1) / DT_TILE_W) * ((R.H + DT_TILE_H - 1) / DT_TILE_H) * 4;
    const float4* bbx = reinterpret_cast<const float4*>(R.blockbox);
    unsigned long long livem = 0ull;                 // objects with a live screen box (usually one or none)
    for (int o = 0; o < m.n_obj; ++o) livem |= s_oboxf[o][0] <= s_oboxf[o][1] ? 1ull << o : 0ull;
    for (int b = tid; b < n_blk; b += DT_OBJSETUP_T) {
      const float4 bb = bbx[b];                      // x0, x1, y0, y1
      unsigned long long mk = 0ull;
      for (unsigned long long lm = livem; lm; lm &= lm - 1ull) {
        const int o = __builtin_ctzll(lm);
        if (!(bb.y < s_oboxf[o][0] || bb.x > s_oboxf[o][1] || bb.w < s_oboxf[o][2] || bb.z > s_oboxf[o][3])) mk |= 1ull << o;
      }
      R.objmask[(size_t)(pos ? pos[e] : e) * n_blk + b] = mk;   // indexed by position in the render order
    }
  }
}


// max(|a|, |b|) in one instruction (source modifiers; no NaN canonicalisation needed: inputs are finite)
__device__ inline float absmax(float a, float b) { float r; asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
// floor(x) as int in one instruction
__device__ inline int flr_i32(float x) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// v_cvt_f32_ubyteN: one instruction per texel channel (the compiler does not pick it)
__device__ inline float ubyte0(uint32_t x) { float r; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ inline float ubyte1(uint32_t x) { float r; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ inline float ubyte2(uint32_t x) { float r; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(x)); return r; }

__device__ inline uint32_t pack_rgb(const float col[3]) {  // glReadPixels float -> unorm8: round(255 c)
  const uint32_t r = (uint32_t)(fminf(fmaxf(col[0], 0.f), 255.f) + 0.5f);
  const uint32_t g = (uint32_t)(fminf(fmaxf(col[1], 0.f), 255.f) + 0.5f);
  const uint32_t b = (uint32_t)(fminf(fmaxf(col[2], 0.f), 255.f) + 0.5f);
  return r | (g << 8) | (b << 16);
}

// ---- generic (exact) per-sample path, used for edge pixels ---------------------------
struct Hit { int cls; int ti, tj; float t, wx, wz; };

__device__ inline void plane_hit(const EnvCam& c, const Ray& r, float h, float& t, float& wx, float& wz) {
  t = h * frcp(-r.yla);
  const float rr = t * r.xe, ff = t * r.fwd;
  wx = c.Cx + rr * c.sa + ff * c.ca;
  wz = c.Cz + rr * c.ca - ff * c.sa;
}

__device__ inline Hit classify(const EnvCam& c, const MapU& m, const TileLds* tiles, const Ray& r) {
  Hit h;
  h.cls = CLS_SKY; h.ti = h.tj = 0; h.t = 0.f; h.wx = h.wz = 0.f;
  if (!(r.yla < 0.f)) return h;
  float t, wx, wz;
  plane_hit(c, r, c.Cy, t, wx, wz);                 // tile plane y = 0
  if (t >= NEAR_Z && t <= FAR_Z) {
    const float fi = floorf(wx * m.its), fj = floorf(wz * m.its);
    if (fi >= 0.f && fj >= 0.f && fi < m.gwf && fj < m.ghf) {
      const int i = (int)fi, j = (int)fj;
      if (tiles[m.tile_off + j * m.gw + i].flags & 1u) {
        h.cls = CLS_TILE; h.ti = i; h.tj = j; h.t = t; h.wx = wx; h.wz = wz;
        return h;
      }
    }
  }
  plane_hit(c, r, c.Cy - GROUND_Y, t, wx, wz);      // ground quad y = -0.008
  if (t >= NEAR_Z && t <= FAR_Z && fabsf(wx) <= GROUND_HALF && fabsf(wz) <= GROUND_HALF) {
    h.cls = CLS_GROUND; h.t = t; h.wx = wx; h.wz = wz;
  }
  return h;
}

// Colour (0..255 floats) of primitive `h` evaluated at the pixel-centre ray `rc`
// (MSAA: coverage per sample, shading once per primitive at the pixel centre).
__device__ inline void shade(const EnvCam& c, const MapU& m, const RenderParams& R, const TileLds* tiles,
                             const Hit& h, const Ray& rc, float out[3]) {
  if (h.cls == CLS_SKY) { out[0] = c.hor[0]; out[1] = c.hor[1]; out[2] = c.hor[2]; return; }
  if (h.cls == CLS_GROUND) {
    float t = h.t, wx = h.wx, wz = h.wz;
    if (rc.yla < 0.f) plane_hit(c, rc, c.Cy - GROUND_Y, t, wx, wz);
    const float ndl = ground_ndl(c, wx, wz);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = c.gnd[k] * fminf(c.base[k] + c.dif[k] * ndl, 1.f);
    return;
  }
  float t = h.t, wx = h.wx, wz = h.wz;
  if (rc.yla < 0.f) plane_hit(c, rc, c.Cy, t, wx, wz);
  const float ndl = plane_ndl(c.L, c.sth, c.cth, rc, t);
  float I[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) I[k] = fminf(c.base[k] + c.dif[k] * ndl, 1.f);
  const float fx = wx * m.its - (float)h.ti, fz = wz * m.its - (float)h.tj;
  tile_color(R, tiles[m.tile_off + h.tj * m.gw + h.ti], fx, fz, I, out);
}


// z-buffer one mesh triangle against the 4 samples of the pixel centred at (pcx, pcy) px.  Depth is kept as
// w = 1 / depth (larger = nearer; 0 = nothing yet): barycentrics and w are affine in the sample position, so a triangle
// costs one set-up at the pixel centre + two fused multiply-adds per quantity and sample, and no division.
template <typename Tri>
__device__ inline void test_tri_inside(const Tri& st, float pcx, float pcy, float wbest[4], int tbest[4]);
template <typename Tri>
__device__ inline void test_tri(const Tri& st, float pcx, float pcy, float wbest[4], int tbest[4]) {
  if (pcx < st.bx0 || pcx > st.bx1 || pcy < st.by0 || pcy > st.by1) return;
  test_tri_inside(st, pcx, pcy, wbest, tbest);
}
// ... the pixel centre is already known to lie in the triangle's (padded) screen box
template <typename Tri>
__device__ inline void test_tri_inside(const Tri& st, float pcx, float pcy, float wbest[4], int tbest[4]) {
  const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
  const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
  // b0(q) = ((x1-qx)(y2-qy) - (x2-qx)(y1-qy)) / area: d/dqx = (y1-y2)/area, d/dqy = (x2-x1)/area; b1 likewise;
  // w(q) = iw2 + b0 (iw0 - iw2) + b1 (iw1 - iw2)
  const float ia = st.inv_area;
  const float e0x = st.sx[0] - pcx, e0y = st.sy[0] - pcy, e1x = st.sx[1] - pcx, e1y = st.sy[1] - pcy, e2x = st.sx[2] - pcx, e2y = st.sy[2] - pcy;
  const float b0c = (e1x * e2y - e2x * e1y) * ia, b1c = (e2x * e0y - e0x * e2y) * ia;
  const float g0x = (e1y - e2y) * ia, g0y = (e2x - e1x) * ia, g1x = (e2y - e0y) * ia, g1y = (e0x - e2x) * ia;
  const float d0 = st.iw[0] - st.iw[2], d1 = st.iw[1] - st.iw[2];
  const float wc = fmaf(b1c, d1, fmaf(b0c, d0, st.iw[2]));
  const float gwx = fmaf(g1x, d1, g0x * d0), gwy = fmaf(g1y, d1, g0y * d0);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float b0 = fmaf(g0y, oy[s], fmaf(g0x, ox[s], b0c));
    const float b1 = fmaf(g1y, oy[s], fmaf(g1x, ox[s], b1c));
    const float b2 = 1.f - b0 - b1;
    const float w = fmaf(gwy, oy[s], fmaf(gwx, ox[s], wc));
    // depth func LESS within [near, far]; equal depth: the earlier triangle in draw order keeps the sample
    const bool in = fminf(fminf(b0, b1), b2) >= 0.f && w <= 1.f / NEAR_Z && w >= 1.f / FAR_Z;
    if (in && (w > wbest[s] || (w == wbest[s] && st.index < tbest[s]))) { wbest[s] = w; tbest[s] = st.index; }
  }
}

// z-buffer the `fill` staged triangles of the wavefront-local LDS chunk against the pixels of the
// lanes with `mine` set.  Two schedules, picked per call from a cost estimate (wave-uniform):
//   pixel-parallel    every `mine` lane collects the staged triangles whose box holds its pixel, then walks its own
//                     list (dense blocks: many pixels);
//   triangle-parallel for each `mine` pixel in turn, the 64 lanes test 64 staged triangles at once and
//                     the per-sample winners are reduced across the wavefront (a handful of pixels of a
//                     small / distant object, where the pixel-parallel loop would idle most lanes).
// Depth func LESS with the draw-order tie break, identical in both schedules.
#ifndef RO_PAIR_CAP
#define RO_PAIR_CAP 256                                  // pair slots per wavefront (16-bit entries)
#endif
#define RO_SCR_BYTES (64 * 4 * 8 + RO_PAIR_CAP * 2)       // per wavefront: sample keys + the pair list
__device__ inline void zbuffer_chunk(const TriCov* w_tris, uint32_t* w_scr, int fill, bool mine, int lane, float pcx, float pcy,
                                     float wbest[4], int tbest[4], int32_t* dbg) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long mm = __ballot(mine);
  const int n_mine = __popcll(mm);
  const int n_chunks = (fill + 63) >> 6;
  const bool tri_parallel = n_mine * (n_chunks * 110 + 110) < fill * 8 + 400;
  if (dbg && lane == 0) {                            // DTSIM_DEBUG_QUEUE statistics
    atomicAdd(dbg + 2, fill); atomicAdd(dbg + 3, n_mine); atomicAdd(dbg + 4, 1); atomicAdd(dbg + 5, tri_parallel ? 1 : 0);
    atomicAdd(reinterpret_cast<unsigned long long*>(dbg + 6), (unsigned long long)fill * (unsigned long long)n_mine);
  }
  if (tri_parallel) {
    unsigned long long todo = mm;
    while (todo) {                                   // wave-uniform
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const float qx = __shfl(pcx, src), qy = __shfl(pcy, src);
      float wb[4] = {0.f, 0.f, 0.f, 0.f};
      int tb[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
      for (int k = lane; k < fill; k += 64) test_tri(w_tris[k], qx, qy, wb, tb);   // tb starts at "no triangle" = +inf: ties keep the lower index
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float wmax = wb[s];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, d));
        int tmin = (wb[s] == wmax && wmax > 0.f) ? tb[s] : 0x7fffffff;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tmin = min(tmin, __shfl_xor(tmin, d));
        if (lane == src && tmin != 0x7fffffff && (wmax > wbest[s] || (wmax == wbest[s] && tmin < tbest[s]))) {
          wbest[s] = wmax; tbest[s] = tmin;
        }
      }
    }
  } else {
    // Pixel-parallel in two steps.  (1) Every lane tests its pixel against the screen boxes of all staged triangles
    // (one broadcast 16-byte LDS read + four compares each) and keeps the hits as a bit mask.  (2) Each lane walks
    // its own mask: the wavefront runs max-over-lanes(candidates) passes of the 4-sample test, with a different
    // triangle per lane.
    static_assert(TRI_CAP <= 128 && TRI_CAP % 32 == 0, "up to four 32-bit candidate masks");
    // Round 3: balanced.  Walking per-lane candidate masks costs max-over-lanes passes of the 4-sample test -- measured
    // 18 passes per call where the (pixel, triangle) pairs would fill 5.8 (profiles/r03_variants_ab.txt block G): the
    // pixels of a batch straddle the dense middle of an object and the empty corners of its box.  Instead (1) the box
    // test of each staged triangle appends its hits to a wavefront-local pair list (ballot + mbcnt: one 16-bit LDS
    // write per hit), (2) the list is drained 64 pairs at a time, every lane testing ITS pair, and the per-sample
    // winners meet in LDS as 64-bit keys (w bits : reversed triangle index) under ds_max_u64 -- LESS depth test with
    // the draw-order tie break, the same order relation as the register compare; (3) each pixel's lane reads its four
    // keys back.
    unsigned long long* zb = reinterpret_cast<unsigned long long*>(w_scr);                   // [64 pixels][4 samples]
    uint16_t* plist = reinterpret_cast<uint16_t*>(w_scr + 64 * 4 * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) zb[lane * 4 + q] = 0ull;
    int base = 0;
#ifdef DT_RO_STATS
    int n_pairs = 0, n_pass = 0;
#endif
    auto drain = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int q0 = 0; q0 < base; q0 += 64) {          // wave-uniform
        const bool act = q0 + lane < base;
        const uint32_t pr = act ? plist[q0 + lane] : 0u;
        const int src = pr & 63;
        const float qx = __shfl(pcx, src), qy = __shfl(pcy, src);
#ifdef DT_RO_STATS
        ++n_pass;
#endif
        if (act) {
          const TriCov& st = w_tris[pr >> 6];
          const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
          const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
          const float ia = st.inv_area;
          const float e0x = st.sx[0] - qx, e0y = st.sy[0] - qy, e1x = st.sx[1] - qx, e1y = st.sy[1] - qy, e2x = st.sx[2] - qx, e2y = st.sy[2] - qy;
          const float b0c = (e1x * e2y - e2x * e1y) * ia, b1c = (e2x * e0y - e0x * e2y) * ia;
          const float g0x = (e1y - e2y) * ia, g0y = (e2x - e1x) * ia, g1x = (e2y - e0y) * ia, g1y = (e0x - e2x) * ia;
          const float d0 = st.iw[0] - st.iw[2], d1 = st.iw[1] - st.iw[2];
          const float wc = fmaf(b1c, d1, fmaf(b0c, d0, st.iw[2]));
          const float gwx = fmaf(g1x, d1, g0x * d0), gwy = fmaf(g1y, d1, g0y * d0);
          const uint32_t rix = 0x7fffffffu - (uint32_t)st.index;
#pragma unroll
          for (int q = 0; q < 4; ++q) {                  // the arithmetic of test_tri_inside, term for term
            const float b0 = fmaf(g0y, oy[q], fmaf(g0x, ox[q], b0c));
            const float b1 = fmaf(g1y, oy[q], fmaf(g1x, ox[q], b1c));
            const float b2 = 1.f - b0 - b1;
            const float w = fmaf(gwy, oy[q], fmaf(gwx, ox[q], wc));
            const bool in = fminf(fminf(b0, b1), b2) >= 0.f && w <= 1.f / NEAR_Z && w >= 1.f / FAR_Z;
            if (in) atomicMax(zb + src * 4 + q, ((unsigned long long)__float_as_uint(w) << 32) | rix);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int jj = 0; jj < fill; ++jj) {                // wave-uniform
      const int j = jj;
      const float4 bb = *reinterpret_cast<const float4*>(&w_tris[j]);              // bx0, bx1, by0, by1
      // four compares straight into scalar masks, ANDed on the scalar unit (as one bool expression the compiler builds the
      // conjunction out of 0 / 1 integers: 17 vector instructions instead of 4)
      const unsigned long long m = mm & __ballot(pcx >= bb.x) & __ballot(pcx <= bb.y) & __ballot(pcy >= bb.z) & __ballot(pcy <= bb.w);
      if (!m) continue;
      if (__builtin_amdgcn_inverse_ballot_w64(m))
        plist[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)((j << 6) | lane);
      base += __popcll(m);
#ifdef DT_RO_STATS
      n_pairs += __popcll(m);
#endif
      if (base > RO_PAIR_CAP - 64) { drain(); base = 0; }
    }
    if (base) drain();
    if (mine) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned long long k = zb[lane * 4 + q];
        const float w = __uint_as_float((uint32_t)(k >> 32));
        const int ix = (int)(0x7fffffffu - (uint32_t)k);
        if (k != 0ull && (w > wbest[q] || (w == wbest[q] && ix < tbest[q]))) { wbest[q] = w; tbest[q] = ix; }
      }
    }
#ifdef DT_RO_STATS
    if (dbg && lane == 0) { atomicAdd(dbg + 8, n_pairs); atomicAdd(dbg + 9, n_pass); atomicAdd(dbg + 10, (n_pairs + 63) >> 6); }
#endif
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// exact 4-sample resolve of one pixel (centre NDC nx, ny): coverage and depth per sample,
// shading once per primitive at the pixel centre; zbest/tbest = 1 / depth and index of the nearest mesh triangle per
// sample (from the mesh pass; tbest < 0: none), z-buffered against the planes here.
template <bool OBJ>
__device__ inline uint32_t shade_msaa(const EnvCam& c, const MapU& m, const RenderParams& R,
                                      const TileLds* tiles, float nx, float ny, const ScreenTri* tris,
                                      const float zbest[4], const int tbest[4]) {
  const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
  const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
  const float sxn = 2.f / (float)R.W, syn = 2.f / (float)R.H;
  const Ray rc = make_ray(nx, ny, c.tx, c.ty, c.sth, c.cth);
  // 1. coverage: which primitive owns each sample.  key: 0 sky, 1 ground, 2|tj<<2|ti<<14 tile,
  //    3|tri<<2 mesh triangle (depth func LESS against the plane hit).
  uint32_t key[4];
  int n_sky = 0, n_gnd = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const Ray rs = make_ray(nx + ox[s] * sxn, ny - oy[s] * syn, c.tx, c.ty, c.sth, c.cth);
    const Hit hs = classify(c, m, tiles, rs);
    const float zplane = hs.cls == CLS_SKY ? 3.0e38f : hs.t;
    if (OBJ && tbest[s] >= 0 && 1.f / zbest[s] < zplane) key[s] = 3u | ((uint32_t)tbest[s] << 2);   // zbest holds w = 1 / depth
    else if (hs.cls == CLS_TILE) key[s] = 2u | ((uint32_t)hs.tj << 2) | ((uint32_t)hs.ti << 14);
    else { key[s] = (uint32_t)hs.cls; n_sky += hs.cls == CLS_SKY; n_gnd += hs.cls == CLS_GROUND; }
  }
#ifdef DT_WAVE_SPANS   // experiment: how many exact-path pixels have all four samples on one primitive (counters behind the span slots)
  if (!OBJ && R.spans) {
    unsigned long long* cn = R.spans + 2 * 2048 * 4 * 8 - 8;
    const bool uni = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
    const unsigned long long all = __ballot(true), ut = __ballot(uni && (key[0] & 3u) == 2u), ug = __ballot(uni && key[0] == 1u), us = __ballot(uni && key[0] == 0u);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(all)) {
      atomicAdd(cn + 0, (unsigned long long)__popcll(all)); atomicAdd(cn + 1, (unsigned long long)__popcll(ut));
      atomicAdd(cn + 2, (unsigned long long)__popcll(ug)); atomicAdd(cn + 3, (unsigned long long)__popcll(us));
    }
  }
#endif
  // 2. shading: once per primitive, at the pixel centre, weighted by its sample count.  Sky and the
  //    ground quad are single primitives; tiles / triangles are walked as a list of distinct keys so
  //    that a wavefront runs max-over-lanes(distinct) passes of the expensive code, not one per sample.
  float acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = (float)n_sky * c.hor[k];
  if (n_gnd > 0) {
    float t = 0.f, wx = 0.f, wz = 0.f;
    plane_hit(c, rc, c.Cy - GROUND_Y, t, wx, wz);    // centre ray extrapolates when it misses (yla >= 0: t < 0)
    if (!(rc.yla < 0.f)) {                           // oracle: keep the sample's own hit in that case
#pragma unroll
      for (int s = 3; s >= 0; --s)
        if (key[s] == 1u) {
          const Ray rs = make_ray(nx + ox[s] * sxn, ny - oy[s] * syn, c.tx, c.ty, c.sth, c.cth);
          plane_hit(c, rs, c.Cy - GROUND_Y, t, wx, wz);
        }
    }
    const float ndl = ground_ndl(c, wx, wz);
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += (float)n_gnd * (c.gnd[k] * fminf(c.base[k] + c.dif[k] * ndl, 1.f));
  }
  uint32_t todo = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) todo |= (key[s] & 3u) >= 2u ? (1u << s) : 0u;
  float tI[3] = {0.f, 0.f, 0.f}, twx = 0.f, twz = 0.f;
  bool have_tile_lit = false;
  while (todo) {
    const int s0 = __builtin_ctz(todo);
    const uint32_t k0 = s0 == 0 ? key[0] : s0 == 1 ? key[1] : s0 == 2 ? key[2] : key[3];
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) { const bool same = ((todo >> s) & 1u) && key[s] == k0; cnt += same; todo &= same ? ~(1u << s) : ~0u; }
    float col[3];
    if ((k0 & 3u) == 2u) {
      if (!have_tile_lit) {                          // tile-plane hit and light of the centre ray: shared by all tiles
        float t = 0.f;
        if (rc.yla < 0.f) plane_hit(c, rc, c.Cy, t, twx, twz);
        else {                                       // centre ray misses the plane: the sample's own hit (first tile sample)
          const int sf = s0;
          const Ray rs = make_ray(nx + ox[sf] * sxn, ny - oy[sf] * syn, c.tx, c.ty, c.sth, c.cth);
          plane_hit(c, rs, c.Cy, t, twx, twz);
        }
        const float ndl = plane_ndl(c.L, c.sth, c.cth, rc, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) tI[k] = fminf(c.base[k] + c.dif[k] * ndl, 1.f);
        have_tile_lit = true;
      }
      const int tj = (int)((k0 >> 2) & 4095u), ti = (int)(k0 >> 14);
      const float fx = twx * m.its - (float)ti, fz = twz * m.its - (float)tj;
      tile_color(R, tiles[m.tile_off + tj * m.gw + ti], fx, fz, tI, col);
    } else if (OBJ) {
      const ScreenTri& st = tris[k0 >> 2];
      const float pcx = (nx + 1.f) * 0.5f * (float)R.W, pcy = (1.f - ny) * 0.5f * (float)R.H;   // pixel centre, px
      const float b0 = ((st.sx[1] - pcx) * (st.sy[2] - pcy) - (st.sx[2] - pcx) * (st.sy[1] - pcy)) * st.inv_area;
      const float b1 = ((st.sx[2] - pcx) * (st.sy[0] - pcy) - (st.sx[0] - pcx) * (st.sy[2] - pcy)) * st.inv_area;
      const float b2 = 1.f - b0 - b1;
      const float inv = 1.f / (b0 * st.iw[0] + b1 * st.iw[1] + b2 * st.iw[2]);
      float tx[3] = {1.f, 1.f, 1.f};                   // GL_MODULATE with the chunk's texture (objmesh.py:360-375)
      if (st.tex >= 0) {
        const float u = (b0 * st.uw[0] + b1 * st.uw[1] + b2 * st.uw[2]) * inv;
        const float v = (b0 * st.vw[0] + b1 * st.vw[1] + b2 * st.vw[2]) * inv;
        mesh_texel(R, st.tex, u == u ? u : 0.f, v == v ? v : 0.f, tx);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float v = (b0 * st.cw[0][k] + b1 * st.cw[1][k] + b2 * st.cw[2][k]) * inv;
        col[k] = fminf(fmaxf(v == v ? v : 0.f, 0.f), 255.f) * tx[k];
      }
    } else { col[0] = col[1] = col[2] = 0.f; }
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += (float)cnt * col[k];
  }
  const float o[3] = {0.25f * acc[0], 0.25f * acc[1], 0.25f * acc[2]};
  return pack_rgb(o);
}

// ---- per-pixel quantities that do not depend on the env when the camera is shared ------
struct PixInv {
  float lr, lf;     // tile-plane hit in the yaw-local frame (right, forward), metres
  float ndl;        // max(0, N.L) of the tile plane at the hit
  float mrg;        // conservative world-space footprint radius of the 4 MSAA samples
  uint32_t flags;   // PF_*
};
#define PF_VALID 1u       // inside the source image (else BORDER_CONSTANT 0)
#define PF_SKY 2u         // all 4 samples certainly miss every plane
#define PF_ALWAYS_EDGE 4u // horizon band / near-far limits: always take the exact path
#define PF_TILE_OK 8u     // tile-plane hit within [near, far]
#define PF_GROUND_OK 16u  // ground-plane hit within [near, far]

__device__ inline PixInv pix_inv(float nx, float ny, bool valid, float tx, float ty, float sth, float cth,
                                 float Cy, const float L[4], float ex_n, float ey_n) {
  PixInv p;
  p.flags = valid ? PF_VALID : 0u;
  p.lr = p.lf = p.ndl = p.mrg = 0.f;
  const Ray r = make_ray(nx, ny, tx, ty, sth, cth);
  const float ex = ex_n * tx, ey = ey_n * ty;
  const float dy = ey * fabsf(cth);
  if (!(r.yla < 0.f)) {
    p.flags |= (r.yla - dy >= 0.f) ? PF_SKY : PF_ALWAYS_EDGE;
    return p;
  }
  const float inv = frcp(-r.yla);
  const float t = Cy * inv;
  p.lr = t * r.xe; p.lf = t * r.fwd;
  p.ndl = plane_ndl(L, sth, cth, r, t);
  const float rho = dy * inv;
  // World-space reach of the MSAA samples around the pixel-centre hit.  A sample offset
  // (dx, dy) px moves the hit by  t*ex*dx  along `right` and, through the change of the ray
  // parameter (kappa = rho/(1-rho) per full y-offset), by (lr, lf)*kappa*dy + t*ey*|sth|*dy along
  // (right, forward).  The rotated-grid samples sit at (+-0.375, +-0.125) and (+-0.125, +-0.375):
  // take the larger of the two reaches (ex/ey already carry the 0.375 and a 1% pad), +5%.
  {
    const float kappa = rho * frcp(fmaxf(1.f - rho, 0.25f));
    const float ax_ = t * ex, ry_ = fabsf(p.lr) * kappa, fy_ = fabsf(p.lf) * kappa + t * ey * fabsf(sth);
    const float third = 1.f / 3.f;
    const float r1x = ax_ + third * ry_, r1f = third * fy_;          // (0.375, 0.125)
    const float r2x = third * ax_ + ry_, r2f = fy_;                  // (0.125, 0.375)
    p.mrg = 1.05f * fsqrt_(fmaxf(r1x * r1x + r1f * r1f, r2x * r2x + r2f * r2f));
  }
  const float tg = (Cy - GROUND_Y) * inv;
  if (t >= NEAR_Z && t <= FAR_Z) p.flags |= PF_TILE_OK;
  if (tg >= NEAR_Z && tg <= FAR_Z) p.flags |= PF_GROUND_OK;
  if (rho > 0.25f || tg * (1.f + 2.f * rho) > FAR_Z * 0.98f || t * (1.f - 2.f * rho) < NEAR_Z * 1.02f)
    p.flags |= PF_ALWAYS_EDGE;
  return p;
}

// Per-env camera (domain randomisation): the same quantities per (pixel, env), with everything that only depends on
// the env folded into DrCam once per env, and the MSAA reach as a cheaper upper bound -- kappa = rho / (1 - rho) <=
// rho + (4/3) rho^2 for rho <= 1/4 (beyond that the pixel takes the exact path anyway), and both sample-offset terms of
// pix_inv are bounded by (ax + ry)^2 + fy^2.  A larger reach only sends more pixels to the exact path.
struct DrCam { float tx, ty, a1, a2, sth, cth, Cy, dy, cex, eys, kgc, ndl_dir; bool directional; };

__device__ inline DrCam dr_cam(const EnvCam& c, float ex_n, float ey_n) {
  DrCam d;
  d.tx = c.tx; d.ty = c.ty; d.a1 = c.ty * c.cth; d.a2 = c.ty * c.sth; d.sth = c.sth; d.cth = c.cth; d.Cy = c.Cy;
  const float ey = ey_n * c.ty;
  d.dy = ey * fabsf(c.cth); d.cex = ex_n * c.tx; d.eys = ey * fabsf(c.sth);
  d.kgc = (c.Cy - GROUND_Y) / c.Cy;
  d.directional = c.L[3] == 0.f;
  d.ndl_dir = fmaxf(c.cth * c.L[1] + c.sth * c.L[2], 0.f);
  return d;
}

__device__ inline PixInv pix_inv_dr(float nx, float ny, bool valid, const DrCam& d, const float L[4]) {
  PixInv p;
  p.flags = valid ? PF_VALID : 0u;
  p.lr = p.lf = p.ndl = p.mrg = 0.f;
  const float xe = nx * d.tx, yla = fmaf(ny, d.a1, -d.sth), fwd = fmaf(ny, d.a2, d.cth);
  if (!(yla < 0.f)) {
    p.flags |= (yla - d.dy >= 0.f) ? PF_SKY : PF_ALWAYS_EDGE;
    return p;
  }
  const float inv = frcp(-yla);
  const float t = d.Cy * inv;
  p.lr = t * xe; p.lf = t * fwd;
  if (d.directional) p.ndl = d.ndl_dir;             // wave-uniform: the DR light is a direction (simulator.py:565-584)
  else { Ray r; r.xe = xe; r.ye = ny * d.ty; r.yla = yla; r.fwd = fwd; p.ndl = plane_ndl(L, d.sth, d.cth, r, t); }
  const float rho = d.dy * inv;
  const float kappa = fmaf(rho * 1.3334f, rho, rho);
  const float sx = fmaf(fabsf(p.lr), kappa, t * d.cex), fy = fmaf(fabsf(p.lf), kappa, t * d.eys);
  p.mrg = 1.05f * fsqrt_(fmaf(sx, sx, fy * fy));
  const float tg = t * d.kgc;
  if (t >= NEAR_Z && t <= FAR_Z) p.flags |= PF_TILE_OK;
  if (tg >= NEAR_Z && tg <= FAR_Z) p.flags |= PF_GROUND_OK;
  if (rho > 0.25f || tg * (1.f + 2.f * rho) > FAR_Z * 0.98f || t * (1.f - 2.f * rho) < NEAR_Z * 1.02f)
    p.flags |= PF_ALWAYS_EDGE;
  return p;
}

// Edge-pixel queue: one fixed region per (workgroup, wavefront) of the raster launch, worst-case
// sized (every pixel of every env of the chunk), so appends need no atomics; entry =
// (env-in-chunk << 8) | (row-slot k << 6 | lane).  k_resolve drains the regions 64 entries at a time.
#define QREGION (WAVE_PIX * ENVS_PER_BLOCK)
// Queue entry: pixel of the wavefront block (bits 0..7) | env position in the chunk << 8 (bits 8..13).  Object-box entries
// (far end of the region, k_resolve_obj's) carry bit 15 when the pixel is ALSO a plane edge, i.e. when what the raster
// stored for it is not final: k_resolve_obj leaves a pixel alone when no mesh triangle covers any of its samples and the
// bit is clear (the raster's one-ray colour is the pixel).  k_raster_v3 / k_raster_v3dr tell; the other rasters always set it.
#define QE_PLANE_EDGE 0x8000u
#define QE_ALWAYS_EDGE 0x4000u    // plane-edge entries of k_raster_v3 (front of the region): never a one-ray pixel (resolve_region skips its interior test)
static_assert(ENVS_PER_BLOCK <= 64, "the env position of a queue entry has six bits");
#define ITEM_B DT_ITEM_B   // 64-entry batches per k_resolve work item
#define ITEMS_PER_WG DT_ITEMS_PER_WG
#define GRAB_MAX 16       // work items per cursor atomic, at most
#ifndef DT_RES_ENVS
#define DT_RES_ENVS 1             // round 4: 8 -> 1 (one env per item: the launch ended on its longest octet; C5 - 8 %, profiles/r04_wave_spans.txt)
#endif
#define RES_ENVS DT_RES_ENVS  // env positions of a chunk per k_resolve_obj work item
static_assert(ENVS_PER_BLOCK % RES_ENVS == 0 && ENVS_PER_BLOCK / RES_ENVS <= ITEMS_PER_WG, "octet items");

// Work items of k_resolve_obj (its own list: R.work[2] = count, [3] = cursor; second part of R.items): one per RES_ENVS
// env positions of a raster workgroup that queued object-box pixels.
// groups: bit g = env group g of the chunk (RES_ENVS positions) has object-box entries in some region of the workgroup; the
// quad-record rasters pass what they queued (round 4: 60 % of the items used to be empty), the others every group.
// Round 4: the list has two ends.  Items of env groups that queued a lot (heavy: bit g, a subset of groups) go to the front in push
// order, the others to the back (from the last slot down); k_resolve_obj walks front to back, so the units of close-up objects -- up to
// 1 024 pixels against every triangle of the object, 100 - 250 us of one wavefront -- start first instead of wherever their raster
// workgroup happened to finish (the launch used to end on a third of the wavefronts, profiles/r04_wave_spans.txt).
// work[2] = front count, work[6] = back count, work[3] = cursor; capacity = one slot per (raster workgroup, env group).
__device__ inline size_t obj_items_cap(const RenderParams& R) {
  const int n_tiles = ((R.W + DT_TILE_W - 1) / DT_TILE_W) * ((R.H + DT_TILE_H - 1) / DT_TILE_H);
  return (size_t)((R.N + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK) * n_tiles * (ENVS_PER_BLOCK / RES_ENVS);
}
typedef unsigned long long envmask_t;                 // one bit per env group of a chunk (ENVS_PER_BLOCK / RES_ENVS <= 64)
__device__ inline void push_obj_items(const RenderParams& R, uint32_t rwg, envmask_t groups = ~0ull, envmask_t heavy = 0ull) {
  const int ng = ENVS_PER_BLOCK / RES_ENVS;
  if (ng < 64) groups &= (1ull << ng) - 1ull;
  heavy &= groups;
  const envmask_t light = groups & ~heavy;
  const int nh = __popcll(heavy), nl = __popcll(light);
  if (nh) {
    int pos = atomicAdd(R.work + 2, nh);
    for (int i = 0; i < ng; ++i) if ((heavy >> i) & 1ull) R.items2[pos++] = rwg * ITEMS_PER_WG + (uint32_t)i;
  }
  if (nl) {
    size_t pos = obj_items_cap(R) - 1 - (size_t)atomicAdd(R.work + 6, nl);
    for (int i = 0; i < ng; ++i) if ((light >> i) & 1ull) R.items2[pos--] = rwg * ITEMS_PER_WG + (uint32_t)i;
  }
}
#ifndef DT_RO_HEAVY
#define DT_RO_HEAVY 128            // entries of one env in one 256-pixel wavefront region from which its unit counts as heavy
#endif
// which env groups of the chunk have entries in THIS wavefront's region: qend_v = the region's fill after each env (lane = position)
__device__ inline envmask_t obj_groups_of(int qend_v, int lane, envmask_t* heavy = nullptr) {
  static_assert(ENVS_PER_BLOCK / RES_ENVS <= 64, "one bit per env group");
  const int up = __shfl_up(qend_v, 1);
  const int cnt = lane < ENVS_PER_BLOCK ? qend_v - (lane == 0 ? 0 : up) : 0;
  const unsigned long long m = __ballot(cnt != 0), mh = __ballot(cnt >= DT_RO_HEAVY);
  envmask_t g = 0ull, gh = 0ull;
#pragma unroll
  for (int i = 0; i < ENVS_PER_BLOCK / RES_ENVS; ++i) {
    g |= ((m >> (i * RES_ENVS)) & ((1ull << RES_ENVS) - 1ull)) ? (1ull << i) : 0ull;
    gh |= ((mh >> (i * RES_ENVS)) & ((1ull << RES_ENVS) - 1ull)) ? (1ull << i) : 0ull;
  }
  if (heavy) *heavy = gh;
  return g;
}

template <bool DR, bool OBJ>
__global__ __launch_bounds__(RB) void k_raster(RenderParams R, const EnvCam* __restrict__ cams,
                                               const EnvFast* __restrict__ fasts, uint8_t* __restrict__ frames, const uint32_t* __restrict__ texels,
                                               const float4* __restrict__ lut, const RenderMapDev* __restrict__ maps,
                                               const TileLds* __restrict__ tile_recs, uint16_t* __restrict__ queue,
                                               int32_t* __restrict__ qcount) {
  extern __shared__ uint32_t s_mem[];
  TileLds* s_tiles = reinterpret_cast<TileLds*>(s_mem);                                   // [n_tile_recs]

  const int tid = threadIdx.x;
  const int npix = R.W * R.H;
  const int tiles_x = (R.W + DT_TILE_W - 1) / DT_TILE_W, n_tiles = tiles_x * ((R.H + DT_TILE_H - 1) / DT_TILE_H);
  const int tile = blockIdx.x % n_tiles;
  const int chunk = blockIdx.x / n_tiles;
  const int e0 = chunk * ENVS_PER_BLOCK;
  const int e1 = min(e0 + ENVS_PER_BLOCK, R.N);
  {  // stage the raster tile records of every map once per workgroup
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tile_recs);
    for (int i = tid; i < R.n_tile_recs * (int)(sizeof(TileLds) / 4); i += RB) s_mem[i] = src[i];
  }
  __syncthreads();

  // Pixel ownership: the workgroup owns a WAVE_W x DT_TILE_H (128 x 8) pixel tile, wavefront w its rows 2w, 2w+1, lane l
  // pixel (k*64 + l) of the wavefront's row-major block (slot k).  Adjacent lanes are adjacent pixels, so the texel
  // addresses of one load instruction are neighbours in the texture, and the 2D footprint of the tile
  // keeps its texture lines in L1.
  const int wave = tid >> 6, lane = tid & 63;
  const int blk = tile * 4 + __builtin_amdgcn_readfirstlane(wave);   // raster wavefront block (objmask / blockbox index)
  const int tile_x0 = (tile % tiles_x) * DT_TILE_W;
  const int wave_y0 = (tile / tiles_x) * DT_TILE_H + wave * (WAVE_PIX / WAVE_W);
  const float aspect = (float)R.W / (float)R.H;
  const float ex_n = 0.375f * 2.f / (float)R.W * 1.01f, ey_n = 0.375f * 2.f / (float)R.H * 1.01f;

  // per-pixel LUT -> registers (shared by all envs)
  float nx[PPT], ny[PPT];
  bool ok[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (tile_x0 + SLOT_X(k, lane) < R.W && wave_y0 + SLOT_Y(k, lane) < R.H) {
      const float4 l = lut[(wave_y0 + SLOT_Y(k, lane)) * R.W + tile_x0 + SLOT_X(k, lane)];
      nx[k] = l.x; ny[k] = l.y; ok[k] = l.z != 0.f;
    } else { nx[k] = ny[k] = 0.f; ok[k] = false; }
  }
  PixInv pv[PPT];
  if (!DR) {
    const CamShared cs = default_cam(aspect);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      pv[k] = pix_inv(nx[k], ny[k], ok[k], cs.tx, cs.ty, cs.sth, cs.cth, cs.Cy, cs.L, ex_n, ey_n);
      pv[k].ndl = fminf(fmaf(cs.dif, pv[k].ndl, cs.base), 1.f);   // lit factor, env-invariant
    }
  }

  // source-pixel bounding box of this wavefront's pixels (for the mesh-object test)
  float spx[PPT], spy[PPT];
  float wbx0 = 1e30f, wbx1 = -1e30f, wby0 = 1e30f, wby1 = -1e30f;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    spx[k] = (nx[k] + 1.f) * 0.5f * (float)R.W; spy[k] = (1.f - ny[k]) * 0.5f * (float)R.H;
    if (ok[k]) { wbx0 = fminf(wbx0, spx[k]); wbx1 = fmaxf(wbx1, spx[k]); wby0 = fminf(wby0, spy[k]); wby1 = fmaxf(wby1, spy[k]); }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    wbx0 = fminf(wbx0, __shfl_xor(wbx0, d)); wbx1 = fmaxf(wbx1, __shfl_xor(wbx1, d));
    wby0 = fminf(wby0, __shfl_xor(wby0, d)); wby1 = fmaxf(wby1, __shfl_xor(wby1, d));
  }

  uint16_t* w_queue = queue + ((size_t)blockIdx.x * (RB / 64) + wave) * QREGION;
  int qn = 0, qo = 0;                                   // wave-uniform queue fill: plane edges (front), object-box pixels (back)
  int qend_v = 0;
  // Frame stores go through a wavefront-private LDS transpose: lane l then owns the 12 bytes of the
  // 4 consecutive pixels 4l..4l+3 (row-major in the wavefront's block) -> three dword stores per lane,
  // 3*WAVE_W contiguous bytes per block row.
  uint32_t* s_px = s_mem + R.n_tile_recs * (sizeof(TileLds) / 4) + wave * WAVE_PIX;
  const int st_x = tile_x0 + (lane * 4) % WAVE_W, st_y = wave_y0 + (lane * 4) / WAVE_W;
  const bool aligned_rows = (R.W & 3) == 0;
  const bool st_ok = aligned_rows && lane * 4 < WAVE_PIX && st_x < R.W && st_y < R.H;
  const size_t st_off = ((size_t)st_y * R.W + st_x) * 3;
  const int tw1 = R.tex_w + 1, xmask = R.tex_w - 1, ymask = R.tex_h - 1;

  // object masks of the chunk's envs for this block: one vector load, lane l <-> env e0 + l (see k_raster_q)
  uint32_t om_lo = 0u, om_hi = 0u;
  if (OBJ && lane < e1 - e0) {
    const unsigned long long v = R.objmask[(size_t)(e0 + lane) * (n_tiles * 4) + blk];
    om_lo = (uint32_t)v; om_hi = (uint32_t)(v >> 32);
  }
  for (int e = e0; e < e1; ++e) {
    const EnvFast f = fasts[e];                      // wave-uniform: one 64-byte scalar load
    float base0 = 0.f, base1 = 0.f, base2 = 0.f, dif0 = 0.f, dif1 = 0.f, dif2 = 0.f;
    if (DR) {
      const EnvCam c = cams[e];
      base0 = c.base[0]; base1 = c.base[1]; base2 = c.base[2]; dif0 = c.dif[0]; dif1 = c.dif[1]; dif2 = c.dif[2];
      const DrCam dc = dr_cam(c, ex_n, ey_n);
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        pv[k] = pix_inv_dr(nx[k], ny[k], ok[k], dc, c.L);
    }
    const uint32_t hor_rgb = f.hor_rgb;
    const float A = f.A, B = f.B, Cxi = f.Cxi, Czi = f.Czi;

    // ---- fast path: one ray per pixel, straight-line (predicated) code so that the LDS
    // tile-record reads and the 8 texel loads of the 4 pixels are all in flight together.
    // Per-pixel predicates are kept as bools (lane masks in SGPR pairs), not as VGPR bit fields.
    uint32_t px[PPT];
    bool edge[PPT], oedge[PPT];                        // oedge: inside a mesh object's screen box (k_resolve_obj's pixels)
#pragma unroll
    for (int k = 0; k < PPT; ++k) oedge[k] = false;
    bool any_work = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) any_work |= (pv[k].flags & (PF_VALID | PF_SKY)) == PF_VALID;
    if (!__ballot(any_work)) {                       // wave-uniform: all sky / border
#pragma unroll
      for (int k = 0; k < PPT; ++k) { px[k] = (pv[k].flags & PF_VALID) ? hor_rgb : 0u; edge[k] = false; }
    } else {
      float fx[PPT], fz[PPT], gxs[PPT], gzs[PPT];
      bool cand[PPT], is_tile[PPT], fast[PPT];
      bool need_ground = false;
      TileLds tr[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const PixInv& p = pv[k];
        const float gx = fmaf(p.lf, B, fmaf(p.lr, A, Cxi));
        const float gz = fmaf(p.lf, -A, fmaf(p.lr, B, Czi));
        gxs[k] = gx; gzs[k] = gz;
        fx[k] = __builtin_amdgcn_fractf(gx); fz[k] = __builtin_amdgcn_fractf(gz);
        const int ti = flr_i32(gx), tj = flr_i32(gz);
        cand[k] = (p.flags & (PF_VALID | PF_SKY | PF_ALWAYS_EDGE)) == PF_VALID;
        const bool ingrid = cand[k] & ((p.flags & PF_TILE_OK) != 0) & ((unsigned)ti < (unsigned)f.gw) & ((unsigned)tj < (unsigned)f.gh);
        const int idx = ingrid ? f.tile_off + (int)__umul24(tj, f.gw) + ti : f.tile_off;
        tr[k] = s_tiles[idx];
        // flags: 0 absent, 1 present (untextured: exact path), 3 present + textured
        is_tile[k] = ingrid & (tr[k].flags != 0u);
        // interior test: every MSAA sample stays inside the tile  <=>  max(|fx-.5|, |fz-.5|) < .5 - mrg/tile_size
        const float dmax = absmax(fx[k] - 0.5f, fz[k] - 0.5f);
        const bool inside = dmax < fmaf(-p.mrg, f.its, 0.5f);
        fast[k] = ingrid & (tr[k].flags == 3u) & inside;       // plain textured tile interior: the one-ray result is exact
        need_ground |= cand[k] & !is_tile[k];
        edge[k] = ((p.flags & (PF_VALID | PF_ALWAYS_EDGE)) == (PF_VALID | PF_ALWAYS_EDGE)) | (is_tile[k] & !fast[k]);
      }
      bool any_fast = false;
#pragma unroll
      for (int k = 0; k < PPT; ++k) any_fast |= fast[k];
      if (!__ballot(any_fast)) {                     // wave-uniform: nothing here takes the textured one-ray result
#pragma unroll                                       // (ground beyond the map, horizon band): no texel loads, no filter
        for (int k = 0; k < PPT; ++k) px[k] = (pv[k].flags & PF_VALID) ? hor_rgb : 0u;
      } else {
      uint2 top2[PPT], bot2[PPT];
      int wxi[PPT], wyi[PPT];                        // 8-bit filter weights (gl_fix)
      const uint8_t* tex_bytes = reinterpret_cast<const uint8_t*>(texels);
      const uint32_t row_bytes = (uint32_t)tw1 * 4u;
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const float x = fmaf(tr[k].mxz, fz[k], fmaf(tr[k].mxx, fx[k], tr[k].ox));
        const float y = fmaf(tr[k].myz, fz[k], fmaf(tr[k].myx, fx[k], tr[k].oy));
        const int xf = gl_fix(x), yf = gl_fix(y);
        wxi[k] = xf & 255; wyi[k] = yf & 255;
        const uint32_t x0 = (uint32_t)((xf >> 8) & xmask), y0 = (uint32_t)((yf >> 8) & ymask);
        // 32-bit byte offset from the (uniform) pool base -> saddr-form loads; always in bounds
        const uint32_t off = (__umul24(y0, (uint32_t)tw1) + x0 + tr[k].tex_off) << 2;
        __builtin_memcpy(&top2[k], tex_bytes + off, 8);
        __builtin_memcpy(&bot2[k], tex_bytes + (off + row_bytes), 8);
      }
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const PixInv& p = pv[k];
        // llvmpipe's integer GL_LINEAR (gl_linear_rgb), then the lit factor: pv.ndl holds min(base + dif*ndl, 1) for the shared
        // camera (all channels equal), max(0, N.L) under domain randomisation (per-channel base / dif)
        int T[3];
        gl_linear_rgb(top2[k].x, top2[k].y, bot2[k].x, bot2[k].y, wxi[k], wyi[k], T);
        float v0 = (float)T[0], v1 = (float)T[1], v2 = (float)T[2];
        if (DR) {
          v0 *= fminf(fmaf(dif0, p.ndl, base0), 1.f);
          v1 *= fminf(fmaf(dif1, p.ndl, base1), 1.f);
          v2 *= fminf(fmaf(dif2, p.ndl, base2), 1.f);
        } else { v0 *= p.ndl; v1 *= p.ndl; v2 *= p.ndl; }
        uint32_t rgb = 0;
        rgb = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0, rgb);
        rgb = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1, rgb);
        rgb = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2, rgb);
        px[k] = fast[k] ? rgb : ((p.flags & PF_VALID) ? hor_rgb : 0u);
      }
      }
      if (__ballot(need_ground)) {                   // wave-uniform: ground quad beyond the map
        const EnvCam c = cams[e];                    // colours / ground-corner light: only needed here
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const PixInv& p = pv[k];
          const bool gcand = cand[k] & !is_tile[k];
          // every sample's tile-plane hit must stay clear of the grid, ground hit inside the quad
          const float wx = gxs[k] * f.ts, wz = gzs[k] * f.ts;
          const float wxg = fmaf(f.kg, wx - f.Cx, f.Cx), wzg = fmaf(f.kg, wz - f.Cz, f.Cz);
          const bool clear = (wx < -p.mrg) | (wx > f.gw_m + p.mrg) | (wz < -p.mrg) | (wz > f.gh_m + p.mrg);
          const bool inq = (fabsf(wxg) + 2.f * p.mrg < GROUND_HALF) & (fabsf(wzg) + 2.f * p.mrg < GROUND_HALF);
          const bool gfast = gcand & clear & inq & ((p.flags & PF_GROUND_OK) != 0);
          const float ndl = ground_ndl(c, wxg, wzg);
          uint32_t rgb = 0;
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[0] * fminf(c.base[0] + c.dif[0] * ndl, 1.f), 0, rgb);
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[1] * fminf(c.base[1] + c.dif[1] * ndl, 1.f), 1, rgb);
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[2] * fminf(c.base[2] + c.dif[2] * ndl, 1.f), 2, rgb);
          px[k] = gfast ? rgb : px[k];
          edge[k] = edge[k] | (gcand & !gfast);
        }
      }
    }

    // ---- mesh objects: every pixel inside the union box of the env's projected triangles
    // takes the exact path (which z-buffers the triangles per sample)
    if (OBJ) {
      unsigned long long om = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)om_hi, e - e0) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)om_lo, e - e0);   // objects whose box meets this block
      const ObjBox* boxes = R.objbox + (size_t)e * DTSIM_MAX_OBJECTS;
      while (om) {
        const ObjBox ob = boxes[__builtin_ctzll(om)];  // wave-uniform
        om &= om - 1ull;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
          if (ok[k] && spx[k] >= ob.bx0 && spx[k] <= ob.bx1 && spy[k] >= ob.by0 && spy[k] <= ob.by1) oedge[k] = true;
      }
    }

    if (aligned_rows) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) s_px[k * 64 + lane] = px[k];
      const uint4 q = *reinterpret_cast<const uint4*>(s_px + (lane * 4) % WAVE_PIX);   // same wavefront: DS ops are ordered
      if (st_ok) {
        uint32_t* d32 = reinterpret_cast<uint32_t*>(frames + (size_t)e * npix * 3 + st_off);   // 12-byte aligned
        // non-temporal: the frame is written once and not read back by this pass (keeps the texels in L2)
        __builtin_nontemporal_store(q.x | (q.y << 24), d32);
        __builtin_nontemporal_store((q.y >> 8) | (q.z << 16), d32 + 1);
        __builtin_nontemporal_store((q.z >> 16) | (q.w << 8), d32 + 2);
      }
    } else {                                           // odd widths: bytes, straight from the owning lane
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        if (tile_x0 + SLOT_X(k, lane) < R.W && wave_y0 + SLOT_Y(k, lane) < R.H) {
          uint8_t* dst = frames + ((size_t)e * npix + (size_t)(wave_y0 + SLOT_Y(k, lane)) * R.W + tile_x0 + SLOT_X(k, lane)) * 3;
          dst[0] = (uint8_t)px[k]; dst[1] = (uint8_t)(px[k] >> 8); dst[2] = (uint8_t)(px[k] >> 16);
        }
    }

    // ---- edge pixels: exact 4-sample resolve, deferred to k_resolve (own launch, own
    // register budget): append them to this wavefront's queue region.
    bool any_edge = false, any_oedge = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) { any_oedge |= oedge[k]; edge[k] &= !oedge[k]; any_edge |= edge[k]; }
    const uint32_t etag = (uint32_t)(e - e0) << 8;
    if (OBJ && !R.no_msaa && __ballot(any_oedge)) {    // wave-uniform: object-box pixels fill the region from its far end
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const bool ek = oedge[k];
        const unsigned long long mk = __ballot(ek);
        if (ek) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
          w_queue[QREGION - 1 - (qo + rank)] = (uint16_t)(etag | QE_PLANE_EDGE | (uint32_t)(k * 64 + lane));   // (this raster does not tell: always "plane edge")
        }
        qo += __popcll(mk);
      }
    }
    if (!R.no_msaa && __ballot(any_edge)) {    // wave-uniform
#pragma unroll
      for (int k = 0; k < PPT; ++k) {                // per pixel slot: ballot -> rank -> masked store
        const bool ek = edge[k];
        const unsigned long long mk = __ballot(ek);
        if (ek) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
          w_queue[qn + rank] = (uint16_t)(etag | (uint32_t)(k * 64 + lane));
        }
        qn += __popcll(mk);
      }
    }
    if (OBJ) qend_v = lane >= e - e0 ? qo : qend_v;   // lane l: object-entry fill after env e0 + l (k_resolve_obj's per-env ranges)
  }
  if (lane == 0) qcount[blockIdx.x * (RB / 64) + wave] = qn;
  if (OBJ && lane < ENVS_PER_BLOCK) R.qend[((size_t)blockIdx.x * (RB / 64) + wave) * ENVS_PER_BLOCK + lane] = (uint16_t)qend_v;
  // Work items for k_resolve: the workgroup's 64-entry batches (four regions, flattened), ITEM_B at a
  // time, appended to a global list so that the resolve launch can spread hot tiles (close-up meshes,
  // horizon band) over the whole chip.  The append order is arbitrary; items touch disjoint pixels.
  __shared__ int s_nb[RB / 64], s_no[RB / 64];
  if (lane == 0) { s_nb[wave] = (qn + 63) >> 6; s_no[wave] = qo; }
  __syncthreads();
  if (tid == 0 && !R.no_msaa) {
    int nb = 0, no = 0;
#pragma unroll
    for (int r = 0; r < RB / 64; ++r) { nb += s_nb[r]; no += s_no[r]; }
    if (nb > 0) {                                    // k_resolve: ITEM_B batches of plane-edge entries per item
      const int ni = (nb + ITEM_B - 1) / ITEM_B;
      const int pos = atomicAdd(R.work, ni);
      for (int i = 0; i < ni; ++i) R.items[pos + i] = (uint32_t)blockIdx.x * ITEMS_PER_WG + (uint32_t)i;
    }
    if (OBJ && no > 0) push_obj_items(R, (uint32_t)blockIdx.x);
  }
}


// ---- per-pixel tables of the shared camera (env-invariant; built by k_pix_setup) ----------------------------------
// PixTab: what k_raster_q keeps in registers per pixel.  lit: < 0 source pixel outside the image (BORDER_CONSTANT 0),
// 0 sky for all four samples, > 0 the lit factor min(base + dif * N.L, 1) of the tile plane at the pixel-centre hit.
// mi: low 16 bits = MSAA reach in quad cells (0xFFFF: never a one-ray pixel), high 16 = the reach in metres, fp16 rounded up.
struct alignas(16) PixTab { float lr, lf, lit; uint32_t mi; };
// SampTab: the four MSAA sample hits on the tile plane in the yaw-local frame (k_resolve_q), as fp16 offsets from the
// pixel-centre hit (lr, lf) of the PixTab (a sample sits within a pixel footprint of the centre: an fp16 offset
// places it to 5e-4 of that footprint; for pixels whose centre ray misses the planes the offsets are the hits
// themselves).  flags: bit s = sample s hits the tile plane within [near, far], bit 4+s = it hits the ground plane
// within [near, far] (ground hit = kg * tile-plane hit).
struct alignas(16) SampTab { uint32_t dlr[2], dlf[2]; uint32_t flags; uint32_t pad[3]; };   // pad: lr, lf, lit of the PixTab (float bits)
static_assert(sizeof(PixTab) == 16 && sizeof(SampTab) == 32, "table records");

__global__ void k_pix_setup(RenderParams R, const float4* __restrict__ lut, PixTab* __restrict__ pixtab, SampTab* __restrict__ samptab) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= R.W * R.H) return;
  const float aspect = (float)R.W / (float)R.H;
  const float ex_n = 0.375f * 2.f / (float)R.W * 1.01f, ey_n = 0.375f * 2.f / (float)R.H * 1.01f;
  const CamShared cs = default_cam(aspect);
  const float4 l = lut[pix];
  const bool ok = l.z != 0.f;
  const PixInv p = pix_inv(l.x, l.y, ok, cs.tx, cs.ty, cs.sth, cs.cth, cs.Cy, cs.L, ex_n, ey_n);
  const bool cand = (p.flags & (PF_VALID | PF_SKY)) == PF_VALID;
  const bool plain = cand && !(p.flags & PF_ALWAYS_EDGE) && (p.flags & PF_TILE_OK) && (p.flags & PF_GROUND_OK);
  PixTab t;
  t.lr = p.lr; t.lf = p.lf;
  t.lit = !ok ? -1.f : !cand ? 0.f : fminf(fmaf(cs.dif, p.ndl, cs.base), 1.f);
  // one-ray pixel  <=>  cells-to-boundary k > reach + 0.5  <=>  k > floor(reach + 0.5)   (k integer)
  t.mi = plain ? (uint32_t)fminf(floorf(fmaf(p.mrg, R.q_per_m, 0.5f)), 65534.f) : 0xFFFFu;
  t.mi |= (uint32_t)__half_as_ushort(__float2half_ru(fminf(p.mrg, 60000.f))) << 16;
  pixtab[pix] = t;
  SampTab sp;
  const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
  const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
  const float sxn = 2.f / (float)R.W, syn = 2.f / (float)R.H;
  sp.flags = 0u;
  float dlr[4], dlf[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Ray r = make_ray(l.x + ox[k] * sxn, l.y - oy[k] * syn, cs.tx, cs.ty, cs.sth, cs.cth);
    dlr[k] = dlf[k] = 0.f;
    if (r.yla < 0.f) {
      const float inv = frcp(-r.yla);
      const float t = cs.Cy * inv, tg = (cs.Cy - GROUND_Y) * inv;
      dlr[k] = fminf(fmaxf(t * r.xe - p.lr, -60000.f), 60000.f); dlf[k] = fminf(fmaxf(t * r.fwd - p.lf, -60000.f), 60000.f);
      if (t >= NEAR_Z && t <= FAR_Z) sp.flags |= 1u << k;
      if (tg >= NEAR_Z && tg <= FAR_Z) sp.flags |= 16u << k;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sp.dlr[j] = (uint32_t)__half_as_ushort(__float2half_rn(dlr[2 * j])) | ((uint32_t)__half_as_ushort(__float2half_rn(dlr[2 * j + 1])) << 16);
    sp.dlf[j] = (uint32_t)__half_as_ushort(__float2half_rn(dlf[2 * j])) | ((uint32_t)__half_as_ushort(__float2half_rn(dlf[2 * j + 1])) << 16);
  }
  sp.pad[0] = __float_as_uint(t.lr); sp.pad[1] = __float_as_uint(t.lf); sp.pad[2] = __float_as_uint(t.lit);   // the PixTab's hit and lit factor once more: phase 2 of the exact path reads ONE table
  samptab[pix] = sp;
}

__device__ inline float med3f(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// ---- resolve_region: exact path of the quad-layout pipeline (shared camera, no mesh objects) ----------------------
// Called by every wavefront of k_raster_q at the end of its env loop on ITS OWN queue region (the entries it appended):
// no second launch, no work list, no atomics -- and the gather-bound resolve of one wavefront overlaps the VALU-bound
// env loops of the other wavefronts on the CU.  Per entry (one pixel of one env):
//   1. the pixel's centre hit -> padded quad coordinates -> its tile's table entry and quad record, as in k_raster_q;
//   2. exact interior test (distance of the hit to the tile boundary, in cells, against the MSAA reach): the
//      cell-granular test of k_raster_q is conservative by up to a cell and sends every pixel of the seam cell here;
//      most entries pass and get the one-ray colour from the same integer filter (bit-identical to k_raster_q);
//   3. otherwise the four samples: their tile-plane hits in the yaw-local frame are env-invariant (SampTab, built by
//      k_pix_setup), so a sample costs one 2x2 transform and one table lookup.  Coverage per sample (tile if the
//      tile plane is hit within [near, far] on a present tile; else the ground quad if hit within range and inside
//      +-50 m; else the clear colour), shading once per distinct primitive at the pixel centre (GL semantics:
//      simulator.py:1932-1934, graphics.py:172-251): a tile is shaded with ITS texture at the centre hit -- outside the
//      tile the coordinate wraps (GL_REPEAT), which the quad records encode -- times the centre's lit factor.
// ---- the quad-record filter ("dtsim8") -------------------------------------------------------------------
// GL's own filter (llvmpipe: gl_linear_rgb above) works with 8-bit weights and 8-bit intermediates, so nothing is gained by filtering
// finer than that; the quad pipeline filters in ONE v_dot4_u32_u8 per channel: the four bilinear weights, times the lit factor
// where it is the same for the three channels (shared camera), times 256, rounded to bytes (v_cvt_pk_u8_f32); colour =
// (sum(texel * weight) + 128) >> 8.  Measured against the GL goldens it differs from llvmpipe by +-1/255 on about a quarter of the
// textured pixels -- what the 16-bit filter it replaces did too (tests/test_gl_golden.py::test_quad_filter_distance_to_gl).
// The texture coordinate itself is snapped to 256ths of a texel first -- llvmpipe rounds its coordinates to 8 fractional bits before it
// splits them into texel index and weight (gl_linear_rgb above) -- by ONE float add: X + 32768 (X in quad cells, 0 <= X < 32768; the host
// checks the padded grid against it) rounds X to a multiple of 1/256 and leaves it in the sum's bit pattern: byte 0 = the fraction in
// 256ths, bits 8..22 = the cell number (S = 256: byte 1 = the cell inside the tile, byte 2 = the tile).  A fraction that rounds up to 1
// carries into the cell number, as in GL.  The hot loops take everything from these bits -- no v_cvt_flr, no v_fract.
#define Q8_SNAP 32768.f
#define Q8_LIT (1.f / 256.f)                         // the weights' light argument is lit / 256 (the fractions come as 0..255)
__device__ inline uint32_t q8_bits(float X) { return __float_as_uint(X + Q8_SNAP); }
__device__ inline uint32_t q8_cell(uint32_t b) { return (b >> 8) & 0x7FFFu; }      // the cell number (one v_bfe_u32)
__device__ inline float q8_frac(uint32_t b) { return (float)(b & 255u); }          // the fraction x 256 (one v_cvt_f32_ubyte0)
// Byte offset of the record of the cell of snapped coordinates (xb, zb) inside a 256 x 256 block, masked by the tile's table entry (0 for the
// one-record blocks): the records lie 4 x 2 cells to a 128-byte line -- offset = (cx >> 2) << 14 | cz << 6 | (cx & 3) << 4 -- because the texture
// unit charges a gather per distinct line, and the footprint of a 32 x 2 pixel slot is a slanted strip: 12.0 lines per gather instead of 14.0 with
// the texture's rows laid end to end (tools/gather_lines_model.py; profiles/r06_variants_ab.txt block H).  cx * 0x1010 puts cx at bits 12.. and
// 4..; cz (byte 1 of zb, shifted to bits 6..13) goes over the middle: 3 instructions + the v_and_or that masks and adds the block's offset.
__device__ inline uint32_t q8_rec256(uint32_t xb, uint32_t zb, uint32_t m) {
  uint32_t P, Zs;                                    // (the byte selects written out: the compiler's SDWA peephole finds them for some of the four pixels only)
  asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(P) : "v"(xb), "s"(0x1010u));
  asm("v_lshlrev_b32_sdwa %0, 6, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(Zs) : "v"(zb));
  return ((P & 0xFC030u) | Zs) & m;
}
// a8, b8: the fractions x 256 (0..255); l = lit / 256 (1 / 256 for an unlit filter).  Byte order = the record's texel order:
// (x0,z0) (x1,z0) (x0,z1) (x1,z1).  (The packed forms in the env loops of k_raster_q / k_raster_v3 are these operations, two pixels each.)
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ inline f2_t fma2(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }   // one v_pk_fma_f32, whatever the contraction mode
__device__ inline uint32_t quad_weights8(float a8, float b8, float l) {
#pragma clang fp contract(off)                       // which products are fused is part of the definition (the oracle restates it): u and the w*1 are rounded products
  const float u = a8 * l, v = fmaf(l, 256.f, -u);
  const float w11 = u * b8, w01 = v * b8, w10 = fmaf(u, 256.f, -w11), w00 = fmaf(v, 256.f, -w01);
  uint32_t W = __builtin_amdgcn_cvt_pk_u8_f32(w00, 0, 0u);
  W = __builtin_amdgcn_cvt_pk_u8_f32(w10, 1, W);
  W = __builtin_amdgcn_cvt_pk_u8_f32(w01, 2, W);
  return __builtin_amdgcn_cvt_pk_u8_f32(w11, 3, W);
}
// one-ray colour 0x00BBGGRR of a record (a8, b8 as above; I = lit factor, 0..1)
__device__ inline uint32_t quad_filter(const uint4& q, float a8, float b8, float I) {
  const uint32_t W = quad_weights8(a8, b8, I * Q8_LIT);
  const uint32_t vr = __builtin_amdgcn_udot4(q.x, W, 128u, false), vg = __builtin_amdgcn_udot4(q.y, W, 128u, false),
                 vb = __builtin_amdgcn_udot4(q.z, W, 128u, false);                 // the channel is byte 1 of each sum
  const uint32_t rg = __builtin_amdgcn_perm(vg, vr, 0x0c0c0501u);
  return __builtin_amdgcn_perm(vb, rg, 0x0c050100u);
}
__device__ inline void quad_filter3(const uint4& q, float a8, float b8, float I, float out[3]) {   // 0..255 floats, unrounded
  const uint32_t W = quad_weights8(a8, b8, I * Q8_LIT);
  out[0] = (float)__builtin_amdgcn_udot4(q.x, W, 0u, false) * (1.f / 256.f);
  out[1] = (float)__builtin_amdgcn_udot4(q.y, W, 0u, false) * (1.f / 256.f);
  out[2] = (float)__builtin_amdgcn_udot4(q.z, W, 0u, false) * (1.f / 256.f);
}

#define RQ_LIST 256                                  // MSAA entries compacted per round (the wavefront's 1 KB of LDS)
// V3: the LDS tile table layout of k_raster_v3 (render_v3.inc: block offset at (tz << 10 | tx << 2) + the map's column
// offset EnvQ.pad[0], the record-offset mask 512 bytes behind it) and its wavefront block shape; (tile_x0, wave_y0) is the
// origin of the wavefront's block either way.
// POOL (k_raster_v3, round 3): the four queue regions of a workgroup are drained as ONE list, an equal share per wavefront.
// w_queue = region 0 of the workgroup, (tile_x0, wave_y0) = origin of the workgroup tile, the wavefront's share = entries
// [i0, i0 + n) of the concatenation, (p1, p2, p3) = where regions 1..3 start in it.  Per wavefront the regions hold anything
// from nothing (sky blocks) to several hundred entries (a seam along the block): pooled, the 64-entry batches run full and
// no wavefront of a workgroup idles while another drains its seam.
template <bool S256, bool V3 = false, bool POOL = false>
__device__ inline void resolve_region(const RenderParams& R, const EnvCam* __restrict__ cams, const EnvQ* __restrict__ envq,
                                      const PixTab* __restrict__ pixtab, const SampTab* __restrict__ samptab,
                                      const uint8_t* __restrict__ qtex, const uint32_t* s_qt, uint32_t* w_list,
                                      const uint16_t* w_queue, const int n, const int e0, const int tile_x0, const int wave_y0,
                                      const int lane, const int i0 = 0, const int p1 = 0, const int p2 = 0, const int p3 = 0,
                                      const uint4* s_envq = nullptr) {
  // s_envq (V3): the EnvQ records of the chunk's 64 positions, staged in LDS by the workgroup -- an entry's constants are three
  // ds_read_b128 instead of three 16-byte gathers through the texture unit (in both phases).
  const int npix = R.W * R.H;
  const int LS = R.qlog2;
  const uint32_t SM = (1u << LS) - 1u;
  const float Sf = (float)(1 << LS), lo = 0.5f * Sf;
  const char* qtb = reinterpret_cast<const char*>(s_qt);
  constexpr int WWc = V3 ? DT_V3_WW : WAVE_W;          // pixel columns of the wavefront block the entries index
  const uint32_t tex_min = 32u;                        // block offsets from here on are textured tiles

  // Table entry (block byte offset, record-offset mask) of the tile that OWNS padded quad coordinates (X, Z): tile
  // boundaries sit at k*S + 0.5 (the GL_LINEAR half-texel shift folded into the coordinates), so ownership is decided
  // on (X - 0.5, Z - 0.5) -- unlike the record lookup, which goes by whole cells.  (ox, oz): the tile's origin.
  auto tile_entry = [&](float X, float Z, const float Xhi, const float Zhi, const uint32_t tab_b, const uint32_t pitch4, uint32_t& ta,
                        float& ox, float& oz) -> uint2 {
    const float Xc = med3f(X - 0.5f, lo, Xhi), Zc = med3f(Z - 0.5f, lo, Zhi);
    const uint32_t ti = (uint32_t)flr_i32(Xc) >> LS, tj = (uint32_t)flr_i32(Zc) >> LS;
    ox = (float)(ti << LS) + 0.5f; oz = (float)(tj << LS) + 0.5f;
    if (V3) {
      ta = (tj << 10) + (ti << 2) + tab_b;
      return make_uint2(*reinterpret_cast<const uint32_t*>(qtb + ta), *reinterpret_cast<const uint32_t*>(qtb + ta + 512));
    }
    ta = (ti << 3) + __umul24(tj, pitch4) + tab_b;
    return *reinterpret_cast<const uint2*>(qtb + ta);
  };
  // quad record of block entry `te` at the cell of the snapped (unclamped) coordinates xb, zb (q8_bits), wrapped into the tile
  auto tile_quad = [&](const uint2 te, const uint32_t xb, const uint32_t zb) -> uint4 {
    if (S256) return *reinterpret_cast<const uint4*>(qtex + (te.x | q8_rec256(xb, zb, te.y)));
    const uint32_t local = (((q8_cell(zb) & SM) << LS) | (q8_cell(xb) & SM)) & te.y;
    return *reinterpret_cast<const uint4*>(qtex + (te.x + (local << 4)));
  };
  auto store_rgb = [&](int e, int pix, uint32_t rgb) {
    uint8_t* dst = R.frames + ((size_t)e * npix + pix) * 3;
#ifdef DT_Q_ABL_NOPATCH
    if (rgb != 0x12345678u) return;                  // ablation: what the byte patches cost (traffic, time)
#endif
    // two stores: a 2-byte aligned half + one byte, whichever way the pixel's 3 bytes fall
    const bool odd = (reinterpret_cast<uintptr_t>(dst) & 1u) != 0u;
    uint8_t* p8 = odd ? dst : dst + 2;
    uint16_t* p16 = reinterpret_cast<uint16_t*>(odd ? dst + 1 : dst);
    *p8 = (uint8_t)(odd ? rgb : rgb >> 16);
    *p16 = (uint16_t)(odd ? rgb >> 8 : rgb);
  };

  // ---- phase 1: per entry, the exact interior test; interior entries get the one-ray colour, the others are
  // compacted into the wavefront's list for phase 2.
  // The (up to four) 64-entry batches of a round go through every stage together, so that each dependent memory round
  // trip (entry -> tables -> tile entry -> quad record -> store) is paid once per round: the phase is latency bound.
  // Instantiated for 1, 2 and 4 batches: most regions hold one seam's worth of entries, not 256.
  auto phase1 = [&](auto utag, const int r0) __attribute__((always_inline)) -> int {
    constexpr int U = decltype(utag)::value;
    int n_list = 0;                                  // wave-uniform
    bool have[U], interior[U], tagged[U], skip[U];
      int pix[U], el[U], env[U];
      PixTab pt[U];
      float Xu[U], Zu[U];
      uint2 te_c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        have[u] = r0 + u * 64 + lane < n;
        int qoff = r0 + u * 64 + lane, rx = 0, ry = 0;
        if (POOL) {                                    // entry of the pooled list -> (region, offset in it)
          constexpr int wx = DT_TILE_W / WWc, rows = WAVE_PIX / WWc;
          const int gi = i0 + qoff;
          const int r = (gi >= p1) + (gi >= p2) + (gi >= p3);
          qoff = r * QREGION + gi - (r == 0 ? 0 : r == 1 ? p1 : r == 2 ? p2 : p3);
          rx = (r % wx) * WWc; ry = (r / wx) * rows;
        }
        // the entries were written by this wavefront (workgroup) a moment ago: bypass the (possibly stale) L1 line
        const uint32_t ent = have[u] ? (uint32_t)__builtin_nontemporal_load(w_queue + qoff) : 0u;
        el[u] = (int)((ent >> 8) & 63u);
        // QE_ALWAYS_EDGE (k_raster_v3, round 4): the pixel can never be a one-ray pixel (horizon band, near / far limits): it goes to
        // the list without the interior test -- and without its loads when the whole batch is such (these entries come in runs)
        tagged[u] = (ent & QE_ALWAYS_EDGE) != 0u;
#ifndef DT_Q_V3_PHASE1                                // (A/B aid: -DDT_Q_V3_PHASE1 restores the interior test for k_raster_v3's entries)
        // k_raster_v3 (round 6): NO interior test -- every entry takes the four samples.  The test resolved 30 % of the entries with one record but cost
        // as much as the four-sample phase (a table gather, a record gather and two patches per entry, latency-bound): without it the exact path is
        // 0.10 ms shorter (profiles/r06_variants_ab.txt block E); an interior pixel's four samples give the one-ray colour anyway.
        if (V3) tagged[u] = true;
#endif
        skip[u] = !__ballot(have[u] && !tagged[u]);    // wave-uniform
        const int lp = (int)(ent & 255u);
        pix[u] = (wave_y0 + ry + lp / WWc) * R.W + tile_x0 + rx + lp % WWc;
      }
      float4 qa[U];
      uint4 qb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qa[u] = make_float4(0.f, 0.f, 0.f, 0.f); qb[u] = make_uint4(0u, 0u, 0u, 0u); env[u] = 0;
        pt[u] = PixTab{0.f, 0.f, 0.f, 0xFFFFu};
        if (skip[u]) continue;                       // wave-uniform
#ifdef DT_ABL_P1_NOPIXTAB                              // ablation (wrong frames): what phase 1's per-entry table gather costs
        pt[u] = PixTab{0.3f + 1e-3f * (float)lane, 0.4f, 0.7f, 0x3c000001u};
#else
        pt[u] = pixtab[pix[u]];
#endif
        uint4 qd;
        if (V3 && s_envq) {
          const uint4* fl = s_envq + el[u] * 4;
          const uint4 a4 = fl[0];
          qa[u] = make_float4(__uint_as_float(a4.x), __uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w));
          qb[u] = fl[1]; qd = fl[3];
        } else {
          const EnvQ* fq = envq + min(e0 + el[u], R.N - 1);   // position in the render order -> constants, frame index
          qa[u] = *reinterpret_cast<const float4*>(&fq->A);     // A, B, Cx, Cz
          qb[u] = *reinterpret_cast<const uint4*>(&fq->Xhi);    // Xhi, Zhi, tab_b, pitch4
          qd = *reinterpret_cast<const uint4*>(&fq->reach);   // reach, env, pad[0], pad[1] (one 16-byte piece: the path is bound by its vector-memory instruction count)
        }
        if (V3) qb[u].z = qd.z;
        env[u] = (int)qd.y;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        interior[u] = false; Xu[u] = Zu[u] = 0.f; te_c[u] = make_uint2(0u, 0u);
        if (skip[u]) continue;                       // wave-uniform
        const float A = qa[u].x, B = qa[u].y, Cx = qa[u].z, Cz = qa[u].w;
        const float Xhi = __uint_as_float(qb[u].x), Zhi = __uint_as_float(qb[u].y);
        Xu[u] = fmaf(pt[u].lf, B, fmaf(pt[u].lr, A, Cx)); Zu[u] = fmaf(pt[u].lf, -A, fmaf(pt[u].lr, B, Cz));
        uint32_t ta_c;
        float ox, oz;
        te_c[u] = tile_entry(Xu[u], Zu[u], Xhi, Zhi, qb[u].z, qb[u].w, ta_c, ox, oz);
        // distance of the hit to the boundary of the tile that owns it, in cells, against the MSAA reach
        const float mrg = __half2float(__ushort_as_half((unsigned short)(pt[u].mi >> 16)));   // metres, rounded up
        const float ux = Xu[u] - ox, uz = Zu[u] - oz;        // in [0, S) inside the owner tile
        const float d = fminf(fminf(ux, Sf - ux), fminf(uz, Sf - uz));
        const bool in_range = Xu[u] - 0.5f >= lo && Xu[u] - 0.5f <= Xhi && Zu[u] - 0.5f >= lo && Zu[u] - 0.5f <= Zhi;
        interior[u] = have[u] && !tagged[u] && in_range && te_c[u].x >= tex_min && pt[u].lit > 0.f && (pt[u].mi & 0xFFFFu) < 0xFFF0u && d > mrg * R.q_per_m;
      }
      uint4 qc[U];
#pragma unroll
#ifdef DT_Q_P1_LOAD_ALL                                // A/B aid (round 5 behaviour): every lane gathers its record, interior or not
      for (int u = 0; u < U; ++u) { qc[u] = make_uint4(0u, 0u, 0u, 0u); if (!skip[u]) qc[u] = tile_quad(te_c[u], q8_bits(Xu[u]), q8_bits(Zu[u])); }   // always in bounds (record 0 / 1 for non-tiles)
#else
      // only the INTERIOR entries (~ 30 %) use the record: the gather runs under their lanes -- its cost on the texture path is per distinct line
      for (int u = 0; u < U; ++u) { qc[u] = make_uint4(0u, 0u, 0u, 0u); if (!skip[u] && interior[u]) qc[u] = tile_quad(te_c[u], q8_bits(Xu[u]), q8_bits(Zu[u])); }
#endif
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!skip[u]) {                                // wave-uniform
          const uint32_t rgb = quad_filter(qc[u], q8_frac(q8_bits(Xu[u])), q8_frac(q8_bits(Zu[u])), pt[u].lit);
#ifdef DT_ABL_P1_NOSTORE
          if (interior[u] && rgb == 0x12345678u) store_rgb(env[u], pix[u], rgb);
#else
          if (interior[u]) store_rgb(env[u], pix[u], rgb);
#endif
        }
        const bool msaa = have[u] && !interior[u];
        const unsigned long long mm = __ballot(msaa);
        if (msaa) w_list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = (uint32_t)pix[u] | ((uint32_t)el[u] << 24);
        n_list += __popcll(mm);
      }
    return n_list;
  };
#ifdef DT_Q_V3_PHASE1
  constexpr bool DIRECT = false;
#else
  constexpr bool DIRECT = V3 && !POOL;                 // k_raster_v3 (round 6): the queue entries ARE the list -- no interior test, no compaction, no LDS round trip
#endif
  for (int r0 = 0; r0 < n; r0 += RQ_LIST) {          // wave-uniform: rounds of up to RQ_LIST entries
    const int rem = n - r0;
    int n_list;
    if constexpr (DIRECT) n_list = min(rem, RQ_LIST);
    else {
      n_list = rem > 128 ? phase1(std::integral_constant<int, 4>{}, r0)
             : rem > 64 ? phase1(std::integral_constant<int, 2>{}, r0) : phase1(std::integral_constant<int, 1>{}, r0);
      // ---- phase 2: the four samples of the listed pixels, on dense lanes
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef DT_Q_ABL_NOPHASE2
    for (int l0 = 0; l0 < 0; l0 += 64) {
#else
    for (int l0 = 0; l0 < n_list; l0 += 64) {          // wave-uniform
#endif
      const bool have = l0 + lane < n_list;
      int pix, el;
      if constexpr (DIRECT) {
        // the entries were written by this wavefront a moment ago: bypass the (possibly stale) L1 line
        const uint32_t ent = have ? (uint32_t)__builtin_nontemporal_load(w_queue + r0 + l0 + lane) : 0u;
        const int lp = (int)(ent & 255u);
        el = (int)((ent >> 8) & 63u);
        pix = have ? (wave_y0 + lp / WWc) * R.W + tile_x0 + lp % WWc : 0;
      } else {
        const uint32_t le = have ? w_list[l0 + lane] : 0u;
        pix = (int)(le & 0xFFFFFFu); el = (int)(le >> 24);
      }
#ifdef DT_ABL_P2_NOSAMPTAB                             // ablation (wrong frames): what phase 2's per-entry table gathers cost
      SampTab sp = samptab[lane];
#else
      const SampTab sp = samptab[pix];
#endif
      PixTab pt;
      if constexpr (V3) { pt.lr = __uint_as_float(sp.pad[0]); pt.lf = __uint_as_float(sp.pad[1]); pt.lit = __uint_as_float(sp.pad[2]); pt.mi = 0u; }   // (one table, two 16-byte loads)
      else pt = pixtab[pix];
      const EnvQ* fq = envq + min(e0 + el, R.N - 1);
      if constexpr (V3) {
        // k_raster_v3's version (round 3): no list of distinct primitives.  The colour of a pixel is the sum over its four
        // samples of the shade of the sample's primitive AT THE PIXEL CENTRE (GL: graphics.py:172-251), so every tile
        // sample adds the integer filter of ITS tile's record at the centre cell (same cell index and weights for every
        // tile: the blocks share one cell grid; a sample that is not on a tile reads the all-zero record 0) to one
        // accumulator per channel -- 6 v_dot4 per sample, exact -- and sky / ground samples add their colour once per
        // count.  Tile look-ups go by v_perm like the env loop's; the +-50 m ground-quad test is made in quad coordinates.
        // the env's constants as three 16-byte pieces (round 5: the exact path is bound by the NUMBER of vector-memory instructions it issues)
        float4 qa;
        uint4 qb, qd;
        if (s_envq) {
          const uint4* fl = s_envq + el * 4;
          const uint4 a4 = fl[0];
          qa = make_float4(__uint_as_float(a4.x), __uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w));
          qb = fl[1]; qd = fl[3];
        } else {
          qa = *reinterpret_cast<const float4*>(&fq->A);
          qb = *reinterpret_cast<const uint4*>(&fq->Xhi); qd = *reinterpret_cast<const uint4*>(&fq->reach);
        }
        const float A = qa.x, B = qa.y, Cx = qa.z, Cz = qa.w, Xhi = __uint_as_float(qb.x), Zhi = __uint_as_float(qb.y);
        const uint32_t tab = qd.z;
        const int e = (int)qd.y;
        const float4* c4 = reinterpret_cast<const float4*>(cams + e);   // EnvCam as 16-byte pieces: [2] = {ty, hor[3]}, [3] = {gnd[3], base0}, [4] = {base1, base2, dif0, dif1}, [5] = {dif2, L..}, [6] = {L3, gndl[0..2]}, [7] = {gndl3, ..}
        const float wCy = default_cam((float)R.W / (float)R.H).Cy;      // shared camera: the same height for every env (what k_cam_setup wrote)
        const float kg = (wCy - GROUND_Y) / wCy;
        const float qpm = __uint_as_float(qd.w);                           // quad cells per metre of the env's map
        const float goff = (float)DT_QRING * Sf + 0.5f, ghalf = GROUND_HALF * qpm;   // world 0 and 50 m in padded quad coordinates
        const float Xu = fmaf(pt.lf, B, fmaf(pt.lr, A, Cx)), Zu = fmaf(pt.lf, -A, fmaf(pt.lr, B, Cz));
        const uint32_t xic = q8_bits(Xu), zic = q8_bits(Zu);              // the centre's snapped coordinates: q8_rec256 works on these bits
        const float lit = pt.lit > 0.f ? pt.lit : 0.55f;
        const uint32_t W8 = quad_weights8(q8_frac(xic), q8_frac(zic), lit * Q8_LIT);
        uint32_t aS[3] = {0u, 0u, 0u};                 // sum over the samples of the byte-weight filter of each sample's record
        int n_sky = 0, n_gnd = 0;
        float gX = 0.f, gZ = 0.f;                      // ground hit (quad coordinates) of the lowest-index ground sample
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        uint32_t raddr_prev = 0u;
#pragma unroll
        for (int s = 3; s >= 0; --s) {
          const uint32_t hr = sp.dlr[s >> 1], hf = sp.dlf[s >> 1];
          const float slr = pt.lr + __half2float(__ushort_as_half((unsigned short)((s & 1) ? hr >> 16 : hr)));
          const float slf = pt.lf + __half2float(__ushort_as_half((unsigned short)((s & 1) ? hf >> 16 : hf)));
          const float Xs = fmaf(slf, B, fmaf(slr, A, Cx)), Zs = fmaf(slf, -A, fmaf(slr, B, Cz));
          // the tile that OWNS the sample (ownership on (X - 0.5, Z - 0.5), clamped into the padded grid)
          const float Xo = Xs - 0.5f, Zo = Zs - 0.5f;
          const float Xc = med3f(Xo, lo, Xhi), Zc = med3f(Zo, lo, Zhi);
          const bool s_in = (Xc == Xo) & (Zc == Zo);
          const uint32_t ta = (__builtin_amdgcn_perm((uint32_t)flr_i32(Zc), (uint32_t)flr_i32(Xc), 0x0c0c0501u) << 2) + tab;
          const uint32_t* tp = reinterpret_cast<const uint32_t*>(qtb + ta);
          const uint32_t tb = tp[0], sel = tp[128];
          const bool is_tile = have & (((sp.flags >> s) & 1u) != 0u) & s_in & (tb != 0u);
          // ground-quad hit of the sample: camera + kg * (tile-plane hit - camera), inside +-50 m
          const float Xg = fmaf(kg, Xs - Cx, Cx), Zg = fmaf(kg, Zs - Cz, Cz);
          const bool is_gnd = have & !is_tile & (((sp.flags >> (4 + s)) & 1u) != 0u) & (fabsf(Xg - goff) <= ghalf) & (fabsf(Zg - goff) <= ghalf);
          n_gnd += is_gnd; n_sky += !(is_tile | is_gnd);
          if (is_gnd) { gX = Xg; gZ = Zg; }
          // one record per DISTINCT tile among the pixel's samples (round 4: the path is texture-unit heavy, its gathers fully divergent;
          // the samples of most edge pixels share a tile, or are not on a tile at all): a lane loads only when its address changes
          const uint32_t raddr = is_tile ? (tb | q8_rec256(xic, zic, sel)) : 0u;
          if (s == 3 || raddr != raddr_prev) rec = *reinterpret_cast<const uint4*>(qtex + raddr);
          raddr_prev = raddr;
          aS[0] = __builtin_amdgcn_udot4(rec.x, W8, aS[0], false);
          aS[1] = __builtin_amdgcn_udot4(rec.y, W8, aS[1], false);
          aS[2] = __builtin_amdgcn_udot4(rec.z, W8, aS[2], false);
        }
        float acc[3];
        float4 hc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (__ballot(n_sky > 0)) hc = c4[2];           // wave-uniform: the horizon colour only where some sample sees the sky
        acc[0] = fmaf((float)n_sky, hc.y, (float)aS[0] * (1.f / 256.f));
        acc[1] = fmaf((float)n_sky, hc.z, (float)aS[1] * (1.f / 256.f));
        acc[2] = fmaf((float)n_sky, hc.w, (float)aS[2] * (1.f / 256.f));
        if (__ballot(n_gnd > 0)) {                     // wave-uniform: shade the ground quad (lit at its corners, bilinear)
          if (pt.lit > 0.f && (pt.lr != 0.f || pt.lf != 0.f)) { gX = fmaf(kg, Xu - Cx, Cx); gZ = fmaf(kg, Zu - Cz, Cz); }   // the centre ray hits the planes
          const float hs = 0.5f / ghalf;
          const float a_ = fminf(fmaxf(fmaf(gX - goff, hs, 0.5f), 0.f), 1.f), b_ = fminf(fmaxf(fmaf(gZ - goff, hs, 0.5f), 0.f), 1.f);
          const float4 q3 = c4[3], q4 = c4[4], q5 = c4[5], q6 = c4[6], q7 = c4[7];
          const float n0 = q6.y + a_ * (q6.z - q6.y), n1 = q6.w + a_ * (q7.x - q6.w);
          const float ndl = n0 + b_ * (n1 - n0);
          const float ng = (float)n_gnd;
          acc[0] += ng * (q3.x * fminf(q3.w + q4.z * ndl, 1.f));
          acc[1] += ng * (q3.y * fminf(q4.x + q4.w * ndl, 1.f));
          acc[2] += ng * (q3.z * fminf(q4.y + q5.x * ndl, 1.f));
        }
        const float o[3] = {0.25f * acc[0], 0.25f * acc[1], 0.25f * acc[2]};
#ifdef DT_ABL_P2_NOSTORE
        if (have && o[0] < -1.f) store_rgb(e, pix, pack_rgb(o));
#else
        if (have) store_rgb(e, pix, pack_rgb(o));
#endif
        continue;
      }
      const float A = fq->A, B = fq->B, Cx = fq->Cx, Cz = fq->Cz, Xhi = fq->Xhi, Zhi = fq->Zhi;
      const uint32_t tab_b = V3 ? fq->pad[0] : fq->tab_b, pitch4 = fq->pitch4;
      const int e = (int)fq->env;
      const EnvCam* c = cams + e;
      const float wCx = c->Cx, wCy = c->Cy, wCz = c->Cz, sa = c->sa, ca = c->ca;
      const float kg = (wCy - GROUND_Y) / wCy;
      const float Xu = fmaf(pt.lf, B, fmaf(pt.lr, A, Cx)), Zu = fmaf(pt.lf, -A, fmaf(pt.lr, B, Cz));
      const uint32_t xub = q8_bits(Xu), zub = q8_bits(Zu);
      const float ax = q8_frac(xub), az = q8_frac(zub);
      const float lit = pt.lit > 0.f ? pt.lit : 0.55f;
      // coverage: per sample tile (tile plane hit within [near, far] on a present tile), else ground quad, else clear colour
      uint32_t key[4];                                 // 0 sky, 1 ground, else 2 + table address of the tile
      uint2 te[4];
      int n_sky = 0, n_gnd = 0;
      float gwx = 0.f, gwz = 0.f;                      // ground hit used for shading: the lowest-index ground sample's
#pragma unroll
      for (int s = 3; s >= 0; --s) {
        const uint32_t hr = sp.dlr[s >> 1], hf = sp.dlf[s >> 1];
        const float slr = pt.lr + __half2float(__ushort_as_half((unsigned short)((s & 1) ? hr >> 16 : hr)));
        const float slf = pt.lf + __half2float(__ushort_as_half((unsigned short)((s & 1) ? hf >> 16 : hf)));
        const float Xs = fmaf(slf, B, fmaf(slr, A, Cx)), Zs = fmaf(slf, -A, fmaf(slr, B, Cz));
        uint32_t ta;
        float sox, soz;
        te[s] = tile_entry(Xs, Zs, Xhi, Zhi, tab_b, pitch4, ta, sox, soz);
        const bool s_in = Xs - 0.5f >= lo && Xs - 0.5f <= Xhi && Zs - 0.5f >= lo && Zs - 0.5f <= Zhi;
        const bool is_tile = ((sp.flags >> s) & 1u) && s_in && te[s].x != 0u;
        // ground-quad hit of the sample (world): camera + kg * (tile-plane hit - camera)
        const float wx = kg * (slr * sa + slf * ca) + wCx, wz = kg * (slr * ca - slf * sa) + wCz;
        const bool is_gnd = !is_tile && ((sp.flags >> (4 + s)) & 1u) && fabsf(wx) <= GROUND_HALF && fabsf(wz) <= GROUND_HALF;
        key[s] = !have ? 0u : is_tile ? 2u + ta : is_gnd ? 1u : 0u;
        n_sky += key[s] == 0u; n_gnd += key[s] == 1u;
        if (is_gnd) { gwx = wx; gwz = wz; }
      }
      // shading, once per primitive at the pixel centre
      float acc[3];
      acc[0] = (float)n_sky * c->hor[0]; acc[1] = (float)n_sky * c->hor[1]; acc[2] = (float)n_sky * c->hor[2];
      if (__ballot(have && n_gnd > 0)) {               // wave-uniform
        if (pt.lit > 0.f && (pt.lr != 0.f || pt.lf != 0.f)) {   // the centre ray hits the planes: shade at its ground hit
          gwx = kg * (pt.lr * sa + pt.lf * ca) + wCx; gwz = kg * (pt.lr * ca - pt.lf * sa) + wCz;
        }
        const float a_ = fminf(fmaxf((gwx + GROUND_HALF) * (0.5f / GROUND_HALF), 0.f), 1.f);
        const float b_ = fminf(fmaxf((gwz + GROUND_HALF) * (0.5f / GROUND_HALF), 0.f), 1.f);
        const float n0 = c->gndl[0] + a_ * (c->gndl[1] - c->gndl[0]), n1 = c->gndl[2] + a_ * (c->gndl[3] - c->gndl[2]);
        const float ndl = n0 + b_ * (n1 - n0);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += (float)n_gnd * (c->gnd[k] * fminf(c->base[k] + c->dif[k] * ndl, 1.f));
      }
      uint32_t todo = 0u;
#pragma unroll
      for (int s = 0; s < 4; ++s) todo |= key[s] >= 2u ? (1u << s) : 0u;
      while (__ballot(todo != 0u)) {                   // wave-uniform: one pass per distinct tile over the lanes
        const int s0 = todo ? __builtin_ctz(todo) : 0;
        const uint32_t k0 = s0 == 0 ? key[0] : s0 == 1 ? key[1] : s0 == 2 ? key[2] : key[3];
        uint2 t0;
        t0.x = s0 == 0 ? te[0].x : s0 == 1 ? te[1].x : s0 == 2 ? te[2].x : te[3].x;
        t0.y = s0 == 0 ? te[0].y : s0 == 1 ? te[1].y : s0 == 2 ? te[2].y : te[3].y;
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) { const bool same = ((todo >> s) & 1u) && key[s] == k0; cnt += same; todo &= same ? ~(1u << s) : ~0u; }
        const uint4 qt = tile_quad(t0, xub, zub);
        float col[3];
        quad_filter3(qt, ax, az, lit, col);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += (float)cnt * col[k];
      }
      const float o[3] = {0.25f * acc[0], 0.25f * acc[1], 0.25f * acc[2]};
      if (have) store_rgb(e, pix, pack_rgb(o));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- quad-layout fast path -------------------------------------------------------------------------------------
// k_raster_q<OBJ>: same work decomposition, queue protocol and store path as k_raster<false, OBJ> (shared camera), with
// the per-pixel cost cut to what the VALU issue rate of gfx950 allows (profiles/r02_ubench_valu_rates.txt: every
// vector opcode costs one ~4-cycle issue slot per wavefront, only v_pk_*_f32 does two results per slot):
//   * the tile lookup and the per-tile affine texture map are gone: tile textures are stored pre-rotated, per
//     (texture, angle), as S x S "quad" records of 16 B -- the four GL_LINEAR taps around one cell, channel-planar,
//     plus a meta dword -- and the hit goes from the yaw-local frame straight to padded quad coordinates (EnvQ);
//     the only table read is the block offset of the cell's tile (LDS, 4 B per tile incl. an off-grid ring);
//   * the bilinear filter is integer, at the precision GL's own filter has (round 6: the byte-weight filter of quad_weights8 /
//     quad_filter -- one v_dot4_u32_u8 per channel against the planar taps, light folded into the weights);
//   * interior / edge classification is one compare: meta.lo = cells to the nearest tile boundary (0 for anything
//     that is not a textured tile) against the pixel's MSAA reach in cells, computed once per pixel (Mi).  Cell 0 of
//     each axis straddles the seam (half of it belongs to the neighbour tile) and is always an edge;
//   * all-sky wavefront blocks (env-invariant with the shared camera) take a three-store loop.
// Anything that is not a fast tile pixel falls into a wave-uniform slow branch that decides ground-fast vs edge and
// appends edge pixels to the wavefront's queue region; the exact path (resolve_region) drains it at the end of the env loop.
// With mesh objects (OBJ) the pixels inside object screen boxes are appended from the far end of the region instead and
// left to k_resolve_obj.

#ifndef DT_Q_WAVES
#define DT_Q_WAVES 5                                 // wavefronts per SIMD the register allocation is held to (96 VGPRs)
#endif
// Frame tiles per group of the launch order: an XCD takes the tiles of a group for one chunk of its slice, then for the next chunk, ..., then the next
// group.  Round 6: HALF the frame per group (150 tiles at 640 x 480; 10 before) -- with 64 envs per workgroup the tiles of ONE chunk fill an XCD, and the
// envs of a chunk look at one region of the map (k_env_sort): one L2 serves one region.  C3 - 4.6 %, C5 - 1.8 %; the whole frame as one group is + 5 ... 12 %
// (profiles/r06_variants_ab.txt block P).  -DDT_Q_TILE_GROUP=n fixes the group size instead (A/B aid).
#ifndef DT_Q_TILE_SPLIT
#define DT_Q_TILE_SPLIT 2
#endif
__host__ __device__ inline int dt_q_tile_group(int n_tiles) {
#ifdef DT_Q_TILE_GROUP
  return DT_Q_TILE_GROUP;
#else
  return (n_tiles + DT_Q_TILE_SPLIT - 1) / DT_Q_TILE_SPLIT;
#endif
}
#ifndef DT_Q_PRIO
#define DT_Q_PRIO 3                                  // s_setprio level while a wavefront issues its quad loads (0: off)
#endif
template <bool OBJ, bool S256>
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(OBJ ? DT_Q_WAVES - 1 : DT_Q_WAVES, OBJ ? DT_Q_WAVES - 1 : DT_Q_WAVES))) void k_raster_q(RenderParams R, const EnvCam* __restrict__ cams, const EnvFast* __restrict__ fasts,
                                                 const EnvQ* __restrict__ envq, uint8_t* __restrict__ frames,
                                                 const uint8_t* __restrict__ qtex, const float4* __restrict__ lut, const PixTab* __restrict__ pixtab, const SampTab* __restrict__ samptab,
                                                 const uint32_t* __restrict__ qtiles, uint16_t* __restrict__ queue,
                                                 int32_t* __restrict__ qcount) {
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_qt = s_mem;                                                                   // [n_qtiles]
  const int tid = threadIdx.x;
  const int npix = R.W * R.H;
  const int tiles_x = (R.W + DT_TILE_W - 1) / DT_TILE_W, n_tiles = tiles_x * ((R.H + DT_TILE_H - 1) / DT_TILE_H);
  // XCD-affine workgroup map: workgroup b runs on XCD b % 8 (round-robin dispatch), and XCD x is given the x-th
  // eighth of the env chunks (all frame tiles of each): with the envs in k_env_sort order, one L2 serves the envs of one
  // region of the map for the whole launch.  The mapping only matters for speed.
  const int n_chunks = (R.N + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK, cpx = (n_chunks + 7) / 8;
  // Within an XCD: groups of dt_q_tile_group() frame tiles, all chunks of the slice for one group before the next group,
  // so that a tile's PixTab slice (16 KB) is read from HBM once per XCD instead of once per chunk, while the workgroups
  // in flight still belong to few chunks.
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int q_tg = dt_q_tile_group(n_tiles);
  const int per_group = q_tg * cpx;
  const int grp = bi / per_group, gi = bi % per_group;
  const int g_tiles = min(q_tg, n_tiles - grp * q_tg);        // the last group may be short
  const int tile = grp * q_tg + gi % g_tiles;
  const int chunk = xcd * cpx + gi / g_tiles;
  if (gi >= g_tiles * cpx || chunk >= n_chunks) return;   // padding workgroups (whole workgroup)
  const int rwg = chunk * n_tiles + tile;            // logical workgroup index: queue regions, counts, work items
  const int e0 = chunk * ENVS_PER_BLOCK;             // positions in the render order
  const int e1 = min(e0 + ENVS_PER_BLOCK, R.N);
  for (int i = tid; i < R.n_qtiles * 2; i += RB) s_qt[i] = qtiles[i];
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int tile_x0 = (tile % tiles_x) * DT_TILE_W;
  const int wave_y0 = (tile / tiles_x) * DT_TILE_H + wave * (WAVE_PIX / WAVE_W);
  const int LS = R.qlog2;
  const uint32_t SM = (1u << LS) - 1u;
  const float lo = 0.5f * (float)(1 << LS);

  // per-pixel invariants (shared camera) from the PixTab: hit in the yaw-local frame, lit factor, MSAA reach, flags
  float lr[PPT], lf[PPT], lit[PPT];
  uint32_t Mi[PPT];
  bool cand[PPT], gok[PPT], valid[PPT];
  uint32_t spxy[PPT];                                 // OBJ: source-pixel coordinates of the slot as two halves (x | y << 16)
  bool any_cand = false;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    PixTab t{0.f, 0.f, -1.f, 0xFFFFu};
    const bool inimg = tile_x0 + SLOT_X(k, lane) < R.W && wave_y0 + SLOT_Y(k, lane) < R.H;
    const int pix = (wave_y0 + SLOT_Y(k, lane)) * R.W + tile_x0 + SLOT_X(k, lane);
    if (inimg) t = pixtab[pix];
    lr[k] = t.lr; lf[k] = t.lf; lit[k] = fmaxf(t.lit, 0.f); Mi[k] = t.mi;
    valid[k] = t.lit >= 0.f; cand[k] = t.lit > 0.f; gok[k] = (t.mi & 0xFFFFu) != 0xFFFFu;
    any_cand |= cand[k];
    spxy[k] = 0u;
    if (OBJ && inimg) {
      const float4 l = lut[pix];
      const float sx = (l.x + 1.f) * 0.5f * (float)R.W, sy = (1.f - l.y) * 0.5f * (float)R.H;
      spxy[k] = (uint32_t)__half_as_ushort(__float2half(sx)) | ((uint32_t)__half_as_ushort(__float2half(sy)) << 16);
    }
  }

  uint16_t* w_queue = queue + ((size_t)rwg * (RB / 64) + wave) * QREGION;
  int qn = 0, qo = 0;                                  // plane-edge entries (front of the region), object-box entries (back)
  int qend_v = 0;                                      // lane l: object-entry fill after the env at position e0 + l
  uint32_t* s_px = s_mem + R.n_qtiles * 2 + wave * WAVE_PIX;
  const int st_x = tile_x0 + (lane * 4) % WAVE_W, st_y = wave_y0 + (lane * 4) / WAVE_W;
  // frame rows are dword aligned (W % 4 == 0: launch precondition; other widths take the generic k_raster)
  const bool st_ok = lane * 4 < WAVE_PIX && st_x < R.W && st_y < R.H;
  const size_t st_off = ((size_t)st_y * R.W + st_x) * 3;

  // object masks (OBJ) are indexed by position in the render order (k_obj_setup)
  const int n_blk = n_tiles * 4, blk = tile * 4 + __builtin_amdgcn_readfirstlane(wave);
  // One vector load up front: lane l holds the mask of position e0 + l (a scalar load per env would sit in the same
  // counter as the LDS reads of the env loop and stall them: measured +0.4 ms); per env two v_readlane.
  static_assert(ENVS_PER_BLOCK <= 64, "one lane per env of the chunk");
  uint32_t om_lo = 0u, om_hi = 0u;
  if (OBJ && lane < e1 - e0) {
    const unsigned long long v = R.objmask[(size_t)(e0 + lane) * n_blk + blk];
    om_lo = (uint32_t)v; om_hi = (uint32_t)(v >> 32);
  }
  auto objmask_of = [&](int e) -> unsigned long long {
    if (!OBJ) return 0ull;
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)om_hi, e - e0) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)om_lo, e - e0);
  };
  // pixels of this block inside the screen boxes of the objects in `om` -> appended from the far end of the region
  auto push_obj = [&](const int e, const uint32_t env, unsigned long long om, bool oedge[PPT]) __attribute__((always_inline)) {
    const ObjBox* boxes = R.objbox + (size_t)env * DTSIM_MAX_OBJECTS;
    // The coordinates are kept as halves (registers); their rounding (<= 0.25 px below 1024) is covered by widening the
    // box: a pixel just outside a box may take the exact path for nothing, one inside always does.
    float sx[PPT], sy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      uint32_t t = spxy[k];
      asm volatile("" : "+v"(t));                    // unpack here, not once ahead of the env loop (8 registers)
      sx[k] = __half2float(__ushort_as_half((unsigned short)(t & 0xFFFFu)));
      sy[k] = __half2float(__ushort_as_half((unsigned short)(t >> 16)));
    }
    while (om) {                                     // wave-uniform
      const ObjBox ob = boxes[__builtin_ctzll(om)];
      om &= om - 1ull;
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        if (valid[k] && sx[k] >= ob.bx0 - 0.3f && sx[k] <= ob.bx1 + 0.3f && sy[k] >= ob.by0 - 0.3f && sy[k] <= ob.by1 + 0.3f) oedge[k] = true;
    }
    bool any_oedge = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) any_oedge |= oedge[k];
    if (__ballot(any_oedge)) {                       // wave-uniform (a pixel is in one list or the other: the region never overflows)
      const uint32_t etag = (uint32_t)(e - e0) << 8;
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const bool ek = oedge[k];
        const unsigned long long mk = __ballot(ek);
        if (ek) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
          w_queue[QREGION - 1 - (qo + rank)] = (uint16_t)(etag | QE_PLANE_EDGE | (uint32_t)(k * 64 + lane));
        }
        qo += __popcll(mk);
      }
    }
  };
  const bool wave_sky = !__ballot(any_cand);
  if (wave_sky) {
    // ---- all sky / border for every env: the lane's 12 bytes are the horizon colour pattern, masked where the
    // source pixel lies outside the rectilinear image.
#pragma unroll
    for (int k = 0; k < PPT; ++k) s_px[k * 64 + lane] = valid[k] ? 0xFFFFFFFFu : 0u;
    const uint4 vm = *reinterpret_cast<const uint4*>(s_px + (lane * 4) % WAVE_PIX);
    const uint32_t m0 = (vm.x & 0x00FFFFFFu) | (vm.y << 24), m1 = ((vm.y >> 8) & 0xFFFFu) | (vm.z << 16), m2 = ((vm.z >> 16) & 0xFFu) | (vm.w << 8);
    for (int e = e0; e < e1; ++e) {
      const EnvQ f = envq[e];
      const unsigned long long om = objmask_of(e);
      if (st_ok) {
        uint32_t* d32 = reinterpret_cast<uint32_t*>(frames + (size_t)f.env * npix * 3 + st_off);
        d32[0] = f.sky[0] & m0; d32[1] = f.sky[1] & m1; d32[2] = f.sky[2] & m2;
      }
      if (OBJ && om) {                               // mesh objects in front of the sky
        bool oedge[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) oedge[k] = false;
        push_obj(e, f.env, om, oedge);
      }
      if (OBJ) qend_v = lane >= e - e0 ? qo : qend_v;
    }
    if (lane == 0) qcount[rwg * (RB / 64) + wave] = OBJ ? qo : 0;
  } else {
  typedef float f2 __attribute__((ext_vector_type(2)));
  static_assert(PPT % 2 == 0, "pixel slots are processed in pairs (packed fp32)");
  f2 lr2[PPT / 2], lf2[PPT / 2], lit2[PPT / 2];
#pragma unroll
  for (int j = 0; j < PPT / 2; ++j) {
    lr2[j] = f2{lr[2 * j], lr[2 * j + 1]}; lf2[j] = f2{lf[2 * j], lf[2 * j + 1]}; lit2[j] = f2{lit[2 * j], lit[2 * j + 1]};
  }
  const char* s_qtb = reinterpret_cast<const char*>(s_qt);
  unsigned long long candm[PPT], validm[PPT], plainm[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) { candm[k] = __ballot(cand[k]); validm[k] = __ballot(valid[k]); plainm[k] = __ballot(gok[k]); }
  uint32_t blk_reach;                                // farthest tile-plane hit of the wavefront's pixels from the camera, cells
  {
    float r2 = 0.f;
#pragma unroll
    for (int k = 0; k < PPT; ++k) r2 = fmaxf(r2, lr[k] * lr[k] + lf[k] * lf[k]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
    const float r = fsqrt_(r2) * R.q_per_m * 1.0001f + 1.f;
    blk_reach = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r < 1.0e9f ? (uint32_t)r : 0x7FFFFFFFu));
  }
  bool any_invalid = false;                          // wave-uniform
#pragma unroll
  for (int k = 0; k < PPT; ++k) any_invalid |= validm[k] != ~0ull;
  // The env loop is software-pipelined two deep: the quad loads of env e+1 are issued BEFORE the frame stores of env e.
  // gfx950 retires vector memory operations of a wavefront in issue order (one vmcnt for loads and stores), so a
  // load issued after a store cannot be waited for without also waiting for that store's write acknowledge; with the
  // loads of the next env ahead of the stores, a wavefront only ever waits for stores that are two envs old.
  struct QStage { uint4 q[PPT]; f2 ax2[PPT / 2], az2[PPT / 2]; };
  auto issue_t = [&](const EnvQ& f, QStage& st, auto clamp_tag) __attribute__((always_inline)) {
    constexpr bool CLAMP = decltype(clamp_tag)::value;
    const f2 vA = f2{f.A, f.A}, vB = f2{f.B, f.B}, vCx = f2{f.Cx, f.Cx}, vCz = f2{f.Cz, f.Cz}, vK = f2{Q8_SNAP, Q8_SNAP};
#pragma unroll
    for (int j = 0; j < PPT / 2; ++j) {
      // two pixels per packed op: X = Cx + lr*A + lf*B, Z = Cz + lr*B - lf*A
      // (+ Q8_SNAP as a third, separate add: ONE rounding to 256ths of a texel; the sum's bits are the fixed-point coordinate, see quad_weights8)
      const f2 X2 = fma2(lf2[j], vB, fma2(lr2[j], vA, vCx)) + vK;       // (explicit fused multiply-adds in resolve_region's order: the snap makes the last bit visible)
      const f2 Z2 = fma2(lf2[j], -vA, fma2(lr2[j], vB, vCz)) + vK;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * j + h;
        float X = h ? X2.y : X2.x, Z = h ? Z2.y : Z2.x;
        if (CLAMP) { X = med3f(X, lo + Q8_SNAP, f.Xhi + Q8_SNAP); Z = med3f(Z, lo + Q8_SNAP, f.Zhi + Q8_SNAP); }
        const uint32_t xb = __float_as_uint(X), zb = __float_as_uint(Z);
        if (h) { st.ax2[j].y = q8_frac(xb); st.az2[j].y = q8_frac(zb); }
        else { st.ax2[j].x = q8_frac(xb); st.az2[j].x = q8_frac(zb); }
        // block offset of the cell's tile: LDS table, byte address = tab_b + (zi >> LS) * pitch4 + ((xi >> LS) << 2)
        // The table entry is 8 bytes: the block's byte offset and how the cell index is formed -- for the two
        // one-record blocks (off-grid, untextured) every cell maps to record 0, so they do not occupy cache lines.
        uint32_t ta, local;
        if (S256) {   // S = 256 and a padded grid under 128 tiles: tile = byte 2, cell = byte 1 of the snapped coordinate
          uint32_t t1, t2;
          asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(t1) : "v"(zb), "s"(f.pitch4));
          asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(t2) : "v"(xb));
          ta = t1 + t2 + f.tab_b;
        } else ta = ((q8_cell(xb) >> LS) << 3) + (__umul24(q8_cell(zb) >> LS, f.pitch4) + f.tab_b);
        const uint2 te = *reinterpret_cast<const uint2*>(s_qtb + ta);
        const uint32_t tb = te.x;
        if (S256) local = q8_rec256(xb, zb, te.y) >> 4;                        // te.y: mask of the record's byte offset (4 x 2 cells per line), or 0
        else local = (((q8_cell(zb) & SM) << LS) | (q8_cell(xb) & SM)) & te.y; // te.y: cell mask
#ifdef DT_Q_NO_LOAD
        st.q[k] = make_uint4(tb + local, tb ^ local, local, 0x00000080u);
#elif 0
        st.q[k] = *reinterpret_cast<const uint4*>(qtex + (tb + ((local & 0x3FFu) << 4)));   // ablation: L1-resident taps
#else
        st.q[k] = *reinterpret_cast<const uint4*>(qtex + (tb + (local << 4)));
#endif
      }
    }
  };
  struct alignas(4) U3 { uint32_t a, b, c; };
  // The lane's 12 bytes of frame e.  The store is unconditional (no branch around it, so that the wait counts the
  // compiler derives for the texel loads do not have to cover it): lanes without pixels write to a dump slot.
  uint8_t* const st_base = st_ok ? frames + st_off : reinterpret_cast<uint8_t*>(R.dump) + lane * 16;
  const uint32_t st_stride = st_ok ? (uint32_t)npix * 3u : 0u;
  auto store = [&](const uint32_t env, const U3& o) __attribute__((always_inline)) {   // env: frame index
    uint32_t* d = reinterpret_cast<uint32_t*>(st_base + (size_t)env * st_stride);
#if defined(DT_Q_NO_STORE)
    if (o.a == 0x12345678u) *d = 1u;
#elif 0
    *reinterpret_cast<U3*>(d) = o;
#else
    // non-temporal: the frame is written once and not read back by this pass; keep the taps in L2
    __builtin_nontemporal_store(o.a, d); __builtin_nontemporal_store(o.b, d + 1); __builtin_nontemporal_store(o.c, d + 2);
#endif
  };
  // Clamping the coordinates into the padded grid (two v_med3 per pixel) is only needed when a hit of this block can
  // leave it: the block's farthest hit (cells, env-invariant) against the camera's distance to the border (per env).
  auto issue = [&](const EnvQ& f, QStage& st) __attribute__((always_inline)) {
#if DT_Q_PRIO
    __builtin_amdgcn_s_setprio(DT_Q_PRIO);           // a wavefront about to issue its quad loads goes first
#endif
    if (f.reach > blk_reach) issue_t(f, st, std::false_type{});       // wave-uniform, scalar compare
    else issue_t(f, st, std::true_type{});
#if DT_Q_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto finish = [&](const int e, const uint32_t env, const uint32_t hor_rgb, const QStage& st, unsigned long long om) __attribute__((always_inline)) -> U3 {   // e: position, env: frame, om: object mask of the block (OBJ)
    U3 out{0u, 0u, 0u};
    uint32_t px[PPT];
    bool edge[PPT], oedge[PPT];                      // oedge: inside a mesh object's screen box (OBJ)
    unsigned long long fastm[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) oedge[k] = false;
    const uint4* q = st.q;
    const f2* ax2 = st.ax2; const f2* az2 = st.az2;
    unsigned long long slow = 0ull;                  // lane masks live in SGPR pairs: predicate logic on the scalar unit
    const uint32_t hor_v = hor_rgb;
#pragma unroll
    for (int j = 0; j < PPT / 2; ++j) {
      if (j) __builtin_amdgcn_sched_barrier(0);      // one pixel pair at a time: fewer live temporaries
      // bilinear weights with the lit factor folded in (quad_weights8's operations: fractions in 256ths, lit / 256), two pixels per packed op
      f2 w00, w10, w01, w11;
      {
#pragma clang fp contract(off)
        const f2 I2 = lit2[j] * Q8_LIT, c256 = f2{256.f, 256.f};
        const f2 u = ax2[j] * I2, v = fma2(I2, c256, -u);
        w11 = u * az2[j]; w01 = v * az2[j];
        w10 = fma2(u, c256, -w11); w00 = fma2(v, c256, -w01);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * j + h;
        const unsigned long long fm = __ballot((q[k].w & 0xFFFFu) > (Mi[k] & 0xFFFFu));
        fastm[k] = fm;
        slow |= candm[k] & ~fm;
        edge[k] = false;
        uint32_t W = __builtin_amdgcn_cvt_pk_u8_f32(h ? w00.y : w00.x, 0, 0u);
        W = __builtin_amdgcn_cvt_pk_u8_f32(h ? w10.y : w10.x, 1, W);
        W = __builtin_amdgcn_cvt_pk_u8_f32(h ? w01.y : w01.x, 2, W);
        W = __builtin_amdgcn_cvt_pk_u8_f32(h ? w11.y : w11.x, 3, W);
        const uint32_t vr = __builtin_amdgcn_udot4(q[k].x, W, 128u, false), vg = __builtin_amdgcn_udot4(q[k].y, W, 128u, false),
                       vb = __builtin_amdgcn_udot4(q[k].z, W, 128u, false);          // the channel is byte 1 of each sum (quad_filter)
        const uint32_t rg = __builtin_amdgcn_perm(vg, vr, 0x0c0c0501u);
        const uint32_t rgb = __builtin_amdgcn_perm(vb, rg, 0x0c050100u);
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(px[k]) : "v"(hor_v), "v"(rgb), "s"(fm));   // not a one-ray tile pixel: clear colour
      }
    }
#ifdef DT_Q_ABL_NOSLOW
    slow = 0ull;
#endif
    if (slow) {                                      // wave-uniform: some pixel is not a certain tile interior
      // Per pixel slot, and only for the slots that have such lanes (scalar tests on lane masks):
      //   off-grid cell: ground quad, if every sample stays off the grid and inside the quad;
      //   anything else: queued for k_resolve -- the seam cell (half of it belongs to the neighbour tile, whose taps
      //     the record does not hold), untextured tiles, the horizon band, and the textured cells the cell-granular
      //     test rejected (conservative by up to a cell; k_resolve redoes it exactly on compacted lanes, which is
      //     cheaper than refining here at a few live lanes per wavefront).
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const unsigned long long sk = candm[k] & ~fastm[k];
        if (!sk) continue;                           // wave-uniform
        const bool sl = (sk >> lane) & 1ull;
        const unsigned long long gm = sk & __ballot((q[k].w >> 16) != 0u) & plainm[k];
        edge[k] = sl;
        if (gm) {
          const float lrk = (k & 1) ? lr2[k / 2].y : lr2[k / 2].x, lfk = (k & 1) ? lf2[k / 2].y : lf2[k / 2].x;
          const float mrgk = __half2float(__ushort_as_half((unsigned short)(Mi[k] >> 16)));
          const EnvFast g = fasts[env];              // scalar loads, only on this (ground beyond the map) path: a load at the
          const EnvCam c = cams[env];                // top of the slow branch would be waited for by every LDS access after it
          const float wx = c.Cx + lrk * c.sa + lfk * c.ca, wz = c.Cz + lrk * c.ca - lfk * c.sa;          // tile-plane hit, world
          const float wxg = fmaf(g.kg, wx - c.Cx, c.Cx), wzg = fmaf(g.kg, wz - c.Cz, c.Cz);              // ground-quad hit
          // every sample's tile-plane hit stays clear of the grid rectangle (an absent tile inside the grid goes to
          // the exact path)
          const bool clear = (wx < -mrgk) | (wx > g.gw_m + mrgk) | (wz < -mrgk) | (wz > g.gh_m + mrgk);
          const bool inq = (fabsf(wxg) + 2.f * mrgk < GROUND_HALF) & (fabsf(wzg) + 2.f * mrgk < GROUND_HALF);
          const bool gfast = ((gm >> lane) & 1ull) & clear & inq;
          const float ndl = ground_ndl(c, wxg, wzg);
          uint32_t rgb = 0;
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[0] * fminf(c.base[0] + c.dif[0] * ndl, 1.f), 0, rgb);
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[1] * fminf(c.base[1] + c.dif[1] * ndl, 1.f), 1, rgb);
          rgb = __builtin_amdgcn_cvt_pk_u8_f32(c.gnd[2] * fminf(c.base[2] + c.dif[2] * ndl, 1.f), 2, rgb);
          px[k] = gfast ? rgb : px[k];
          edge[k] = sl & !gfast;
        }
      }
    }

    if (OBJ && om) push_obj(e, env, om, oedge);   // wave-uniform: some object's screen box meets this block

    if (any_invalid) {                               // wave-uniform, rare: source pixels outside the rectilinear image are 0
#pragma unroll
      for (int k = 0; k < PPT; ++k) px[k] = ((validm[k] >> lane) & 1ull) ? px[k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) s_px[k * 64 + lane] = px[k];
    const uint4 o = *reinterpret_cast<const uint4*>(s_px + (lane * 4) % WAVE_PIX);
    // 4 pixels 0x00BBGGRR -> 12 bytes: three byte permutes
    out = U3{__builtin_amdgcn_perm(o.y, o.x, 0x04020100u), __builtin_amdgcn_perm(o.z, o.y, 0x05040201u), __builtin_amdgcn_perm(o.w, o.z, 0x06050402u)};

    bool any_edge = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) { edge[k] &= !oedge[k]; any_edge |= edge[k]; }
    const uint32_t etag = (uint32_t)(e - e0) << 8;
    if (__ballot(any_edge)) {                          // wave-uniform
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const bool ek = edge[k];
        const unsigned long long mk = __ballot(ek);
        if (ek) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
          w_queue[qn + rank] = (uint16_t)(etag | (uint32_t)(k * 64 + lane));
        }
        qn += __popcll(mk);
      }
    }
    return out;
  };
  {  // one stage, deferred store: the 12 bytes of env e-1 are held in registers and stored right after the loads of
     // env e have been issued, so that waiting for those loads never waits for a store of the same iteration.
     // No branch between the loads and the store (first env peeled).
    QStage sa;
    EnvQ f = envq[e0];
    issue(f, sa);
    uint32_t hor = f.hor_rgb, env = f.env, env_prev;
    f = envq[min(e0 + 1, e1 - 1)];
    U3 held = finish(e0, env, hor, sa, objmask_of(e0));
    if (OBJ) qend_v = qo;
    for (int e = e0 + 1; e < e1; ++e) {
      env_prev = env; hor = f.hor_rgb; env = f.env;
      issue(f, sa);
      store(env_prev, held);
      f = envq[min(e + 1, e1 - 1)];                  // next env's constants: the scalar load has the whole filter to land
      held = finish(e, env, hor, sa, objmask_of(e));
      if (OBJ) qend_v = lane >= e - e0 ? qo : qend_v;
    }
    store(env, held);
  }
  if (lane == 0) qcount[rwg * (RB / 64) + wave] = OBJ ? qo : qn;
#ifdef DT_Q_ABL_NORESOLVE
  if (qn < 0) {
#else
  if (qn > 0) {
#endif
    // exact path for this wavefront's own edge pixels, right here (the frame stores of the env loop are ordered
    // before the byte patches: same wavefront, same addresses)
    __builtin_amdgcn_s_waitcnt(0);                 // queue stores have left the wavefront
    resolve_region<S256>(R, cams, envq, pixtab, samptab, qtex, s_qt, s_px, w_queue, qn, e0, tile_x0, wave_y0, lane);
  }
  }
  if (OBJ) {   // mesh objects: k_resolve_obj drains the object-box entries (back of the regions) -- work items for it
    if (lane < ENVS_PER_BLOCK) R.qend[((size_t)rwg * (RB / 64) + wave) * ENVS_PER_BLOCK + lane] = (uint16_t)qend_v;
    __shared__ int s_nb[RB / 64];
    if (lane == 0) s_nb[wave] = qo;
    __syncthreads();
    if (tid == 0) {
      int nb = 0;
#pragma unroll
      for (int r = 0; r < RB / 64; ++r) nb += s_nb[r];
      if (nb > 0) push_obj_items(R, (uint32_t)rwg);
    }
  }
}

#include "render_v3.inc"
#include "render_v3dr.inc"

// Exact 4-sample resolve of the queued edge pixels, stream-ordered after k_raster so the byte patches
// land after the fast-path stores.  Persistent wavefronts pull work items (ITEM_B consecutive 64-entry
// batches of one raster workgroup's four queue regions) from the global list k_raster appended to.
#ifndef DT_RES_WAVES
#define DT_RES_WAVES 5             // wavefronts per SIMD k_resolve is compiled for (90 VGPRs)
#endif
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(DT_RES_WAVES, DT_RES_WAVES))) void k_resolve(RenderParams R, const EnvCam* __restrict__ cams,
                                                const uint16_t* __restrict__ queue, const int32_t* __restrict__ qcount) {
  extern __shared__ uint32_t s_mem[];
  TileLds* s_tiles = reinterpret_cast<TileLds*>(s_mem);
  const int tid = threadIdx.x;
  const int npix = R.W * R.H;
  const int tiles_x = (R.W + DT_TILE_W - 1) / DT_TILE_W, n_tiles = tiles_x * ((R.H + DT_TILE_H - 1) / DT_TILE_H);
  const int wave = tid >> 6, lane = tid & 63;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(R.tile_recs);
    for (int i = tid; i < R.n_tile_recs * (int)(sizeof(TileLds) / 4); i += RB) s_mem[i] = src[i];
  }
  __syncthreads();
  uint32_t* s_wave = s_mem + R.n_tile_recs * (sizeof(TileLds) / 4);
  EnvCam* w_cams = reinterpret_cast<EnvCam*>(s_wave) + wave * ENVS_PER_BLOCK;                    // wavefront-local
  const int n_items = R.work[0];                     // written by the raster launch
#ifdef DT_WAVE_SPANS   // experiment: wall-clock span of every persistent wavefront (100 MHz ticks), read back by dtsim_render
  const unsigned long long span_t0 = wall_clock64(); unsigned long long span_long = 0, span_first = 0, span_first_at = 0, span_sum = 0, span_last = 0; int span_n = 0;
#endif
  // One atomic buys `grab` items, taken with stride n_grabs through the list: same-address atomics are
  // serialised by the L2 (~0.2 ms per 100 k of them), and the stride keeps the consecutive items of one hot
  // raster workgroup on different wavefronts.  Granularity: ~4 grabs per resident wavefront, <= GRAB_MAX items.
  const int grab = max(1, min(GRAB_MAX, n_items / (int)(gridDim.x * (RB / 64) * 4)));
  const int n_grabs = (n_items + grab - 1) / grab;
  // Round 4: the first grab of a wavefront is its own index (the launch holds exactly the resident workgroups, resident_blocks()):
  // 5 000 wavefronts starting on one cursor atomic spent 30 - 60 us each waiting for it (profiles/r04_wave_spans.txt).
  const int n_waves = (int)gridDim.x * (RB / 64);
  bool first_grab = true;
  while (true) {
    int g = wave * (int)gridDim.x + (int)blockIdx.x;
    if (!first_grab) {
      if (lane == 0) g = n_waves + atomicAdd(R.work + 1, 1);
      g = __builtin_amdgcn_readfirstlane(g);
    }
    first_grab = false;
    if (g >= n_grabs) break;
    for (int it = g; it < n_items; it += n_grabs) {  // wave-uniform
#ifdef DT_WAVE_SPANS
      const unsigned long long span_i0 = wall_clock64(); ++span_n;
#endif
      const uint32_t item = R.items[it];
      const int rwg = (int)(item / ITEMS_PER_WG), part = (int)(item % ITEMS_PER_WG);
      const int tile = rwg % n_tiles, chunk = rwg / n_tiles;
      const int e0 = chunk * ENVS_PER_BLOCK;
      {  // the chunk's EnvCams -> wavefront-private LDS (64 bytes per lane; DS ops of one wavefront are ordered)
        static_assert((ENVS_PER_BLOCK * sizeof(EnvCam)) % (64 * 64) == 0, "whole 64-byte slices per lane");
        constexpr int SL = (int)(ENVS_PER_BLOCK * sizeof(EnvCam) / (64 * 64));   // 64-byte slices per lane
        const int ne = min(ENVS_PER_BLOCK, R.N - e0);
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
          const int o = (sl * 64 + lane) * 4;          // uint4 index of the slice
          const uint4* src = reinterpret_cast<const uint4*>(cams + e0) + o;
          uint4* dst = reinterpret_cast<uint4*>(w_cams) + o;
          if (o * 16 < ne * (int)sizeof(EnvCam)) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; }
        }
      }
      const int tile_x0 = (tile % tiles_x) * DT_TILE_W, tile_y0 = (tile / tiles_x) * DT_TILE_H;
      static_assert(RB / 64 == 4, "region lookup assumes 4 wavefronts");
      int rn[4], rb[5];
      rb[0] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) { rn[r] = qcount[rwg * 4 + r]; rb[r + 1] = rb[r] + ((rn[r] + 63) >> 6); }
      const int b_end = min(rb[4], (part + 1) * ITEM_B);
      for (int b = part * ITEM_B; b < b_end; ++b) {  // wave-uniform
        const int reg = (b >= rb[1]) + (b >= rb[2]) + (b >= rb[3]);
        const int q0 = (b - (reg == 0 ? rb[0] : reg == 1 ? rb[1] : reg == 2 ? rb[2] : rb[3])) * 64;
        const int n = reg == 0 ? rn[0] : reg == 1 ? rn[1] : reg == 2 ? rn[2] : rn[3];
        const uint16_t* w_queue = queue + ((size_t)rwg * 4 + reg) * QREGION;
        const int wave_y0 = tile_y0 + reg * (WAVE_PIX / WAVE_W);
        const bool have = q0 + lane < n;
        const uint32_t ent = have ? w_queue[q0 + lane] : 0u;
        const int el = have ? (int)(ent >> 8) : -1, lp = ent & 255;
        const int pix = (wave_y0 + lp / WAVE_W) * R.W + tile_x0 + lp % WAVE_W;      // entries only exist for in-image pixels
        const float4 l = reinterpret_cast<const float4*>(R.lut)[pix];
        if (have) {
          const float wb[4] = {0.f, 0.f, 0.f, 0.f};
          const int tb[4] = {-1, -1, -1, -1};
          const EnvCam c = w_cams[el];
          const MapU m = map_u(R.maps[c.map_id]);
          const uint32_t v = shade_msaa<false>(c, m, R, s_tiles, l.x, l.y, nullptr, wb, tb);
          uint8_t* dst = R.frames + ((size_t)(e0 + el) * npix + pix) * 3;
          dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
        }
      }
#ifdef DT_WAVE_SPANS
      { const unsigned long long d = wall_clock64() - span_i0; span_long = d > span_long ? d : span_long; span_sum += d; span_last = d; if (span_n == 1) { span_first = d; span_first_at = span_i0 - span_t0; } }
#endif
    }
  }
#ifdef DT_WAVE_SPANS
  if (R.spans && lane == 0) {
    unsigned long long* o = R.spans + ((size_t)blockIdx.x * 4 + wave) * 8;
    o[0] = span_t0; o[1] = wall_clock64(); o[2] = (unsigned long long)span_n; o[3] = span_long; o[4] = span_first; o[5] = span_first_at; o[6] = span_sum; o[7] = span_last;
  }
#endif
}

// Exact path of the pixels inside mesh-object screen boxes (and, after the generic raster, of every queued pixel).
// Work unit = (raster tile, env): the entries one env left in the four wavefront regions of a raster workgroup are taken
// together, up to NB 64-entry batches at a time, so that the env's triangles are streamed, culled and staged ONCE for
// the 128 x 8 pixels of the tile instead of once per 128 x 2 block (the round-1 kernel: one pass per
// (batch, env) pair, ~45 pairs per env and frame).  The raster records the queue fill of every region after every env
// (qend), so a unit's entries are four contiguous ranges; a work item covers RES_ENVS consecutive env positions of a chunk.
#ifndef DT_RES_NB
#define DT_RES_NB 2
#endif
#ifndef DT_RO_WAVES
#define DT_RO_WAVES 4              // wavefronts per SIMD the kernel is compiled for (142 VGPRs at 3; 4 caps it at 128)
#endif
template <int NB>
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(DT_RO_WAVES, DT_RO_WAVES))) void k_resolve_obj(RenderParams R, const EnvCam* __restrict__ cams, const uint16_t* __restrict__ queue,
                                                    const int back, const EnvQ* __restrict__ envq) {
  extern __shared__ uint32_t s_mem[];
  TileLds* s_tiles = reinterpret_cast<TileLds*>(s_mem);
  const int tid = threadIdx.x;
  const int npix = R.W * R.H;
  const int tiles_x = (R.W + DT_TILE_W - 1) / DT_TILE_W, n_tiles = tiles_x * ((R.H + DT_TILE_H - 1) / DT_TILE_H);
  const int wave = tid >> 6, lane = tid & 63;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(R.tile_recs);
    for (int i = tid; i < R.n_tile_recs * (int)(sizeof(TileLds) / 4); i += RB) s_mem[i] = src[i];
  }
  __syncthreads();
  uint32_t* s_wave = s_mem + R.n_tile_recs * (sizeof(TileLds) / 4);
  static_assert(RES_ENVS == 1, "a work item is ONE (raster tile, env) unit: everything about it is wave-uniform");
  TriCov* w_tris = reinterpret_cast<TriCov*>(s_wave) + wave * TRI_CAP;                             // wavefront-local
  uint32_t* w_scr = reinterpret_cast<uint32_t*>(reinterpret_cast<TriCov*>(s_wave) + (RB / 64) * TRI_CAP) + wave * (RO_SCR_BYTES / 4);   // z-buffer scratch of the pair schedule
  const int n_front = R.work[2], n_items = n_front + R.work[6];   // written by the raster launch (push_obj_items): heavy items first
  const size_t items_cap = obj_items_cap(R);
  auto item_at = [&](int i) -> uint32_t { return R.items2[i < n_front ? (size_t)i : items_cap - 1 - (size_t)(i - n_front)]; };
#ifdef DT_WAVE_SPANS
  const unsigned long long span_t0 = wall_clock64(); unsigned long long span_long = 0, span_first = 0, span_first_at = 0, span_sum = 0, span_last = 0; int span_n = 0;
#endif
  const int grab = max(1, min(GRAB_MAX, n_items / (int)(gridDim.x * (RB / 64) * 4)));
  const int n_grabs = (n_items + grab - 1) / grab;
  const int n_waves = (int)gridDim.x * (RB / 64);     // first grab = the wavefront's own index, as in k_resolve
  bool first_grab = true;
  while (true) {
    int g = wave * (int)gridDim.x + (int)blockIdx.x;
    if (!first_grab) {
      if (lane == 0) g = n_waves + atomicAdd(R.work + 3, 1);
      g = __builtin_amdgcn_readfirstlane(g);
    }
    first_grab = false;
    if (g >= n_grabs) break;
    // The item loop is software-pipelined: everything a unit needs before it can touch its entries -- the id of the item after next, and for
    // the NEXT item its four entry ranges (the fills of the regions after env p and after env p - 1: eight 16-bit loads on eight lanes), the
    // frame index of its position and the object masks of its four blocks -- is loaded while the current unit is processed.  Round 5: a
    // unit is one env, so all of this is wave-uniform; the per-item staging of 32 envs' fills, 32 envs' masks and the EnvCam copy through
    // LDS (with its barrier) are gone: that chain of dependent round trips was ~ 280 of the kernel's 600 us (profiles/r05_variants_ab.txt, block H).
    static_assert(RB / 64 == 4, "four regions per raster workgroup");
    auto prefetch = [&](const uint32_t item_, uint32_t& qv, int& ev, unsigned long long& hv) {
      const int rwg_ = (int)(item_ / ITEMS_PER_WG), p_ = (int)(item_ % ITEMS_PER_WG);
      const int tile_ = rwg_ % n_tiles, pos_ = min((rwg_ / n_tiles) * ENVS_PER_BLOCK + p_, R.N - 1);
      qv = 0u;                                          // lane 2r: fill of region r after env p, lane 2r + 1: after env p - 1 (= where p's entries start)
      if (lane < 8) { const int pp = p_ - (lane & 1); if (pp >= 0) qv = R.qend[((size_t)rwg_ * 4 + (lane >> 1)) * ENVS_PER_BLOCK + pp]; }
      ev = envq ? (int)envq[pos_].env : pos_;           // position in the render order -> env (one address for the wavefront)
      hv = 0ull;
      if (lane < 4) hv = R.objmask[((size_t)pos_ * n_tiles + tile_) * 4 + lane];
    };
    uint32_t item_cur = item_at(g), item_nxt = g + n_grabs < n_items ? item_at(g + n_grabs) : 0u;
    uint32_t qv_nxt; int ev_nxt; unsigned long long hv_nxt;
    prefetch(item_cur, qv_nxt, ev_nxt, hv_nxt);
    for (int it = g; it < n_items; it += n_grabs) {  // wave-uniform
      const uint32_t item = item_cur;
      const uint32_t qv = qv_nxt;
      const int ev = ev_nxt;
      const unsigned long long hv = hv_nxt;
      item_cur = item_nxt;
      if (it + n_grabs < n_items) prefetch(item_cur, qv_nxt, ev_nxt, hv_nxt);
      item_nxt = it + 2 * n_grabs < n_items ? item_at(it + 2 * n_grabs) : 0u;
      const int rwg = (int)(item / ITEMS_PER_WG), p0 = (int)(item % ITEMS_PER_WG);
      const int tile = rwg % n_tiles, chunk = rwg / n_tiles;
      const int e0 = chunk * ENVS_PER_BLOCK;
      const int ne = min(ENVS_PER_BLOCK, R.N - e0);
      if (p0 >= ne) continue;
      const int tile_x0 = (tile % tiles_x) * DT_TILE_W, tile_y0 = (tile / tiles_x) * DT_TILE_H;
#ifdef DT_WAVE_SPANS
      const unsigned long long span_i0 = wall_clock64(); ++span_n;
#endif
      {                                                // the (tile, env) unit
        const int p = p0;
        int c_[4], s_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_[r] = __builtin_amdgcn_readlane((int)qv, 2 * r + 1); c_[r] = __builtin_amdgcn_readlane((int)qv, 2 * r) - s_[r]; }
        const int t1 = c_[0], t2 = t1 + c_[1], t3 = t2 + c_[2], n_p = t3 + c_[3];
        if (n_p == 0) continue;
        unsigned long long hm0 = 0ull;                 // OR of the four blocks' masks: the unit spans the whole raster tile
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hm0 |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hv >> 32), r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hv, r);
        const int env_p = __builtin_amdgcn_readfirstlane(ev);
        const EnvCam& c = cams[env_p];                 // wave-uniform address: scalar loads where the fields are used
        (void)p;
        const MapU m = map_u(R.maps[c.map_id]);
        const ScreenTri* base = R.stris + (size_t)env_p * R.max_tris;
        const uint2* rng = R.objrange + (size_t)__builtin_amdgcn_readfirstlane(c.map_id) * DTSIM_MAX_OBJECTS;
        for (int g0 = 0; g0 < n_p; g0 += 64 * NB) {   // wave-uniform: up to NB batches of the unit at a time
          bool have[NB], pedge[NB];
          int pix[NB];
          float nxv[NB], nyv[NB];
          float zbest[NB][4];
          int tbest[NB][4];
          float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int idx = g0 + j * 64 + lane;
            have[j] = idx < n_p;
            const int r = (idx >= t1) + (idx >= t2) + (idx >= t3);
            const int li = idx - (r == 0 ? 0 : r == 1 ? t1 : r == 2 ? t2 : t3) + (r == 0 ? s_[0] : r == 1 ? s_[1] : r == 2 ? s_[2] : s_[3]);
            const uint16_t* w_queue = queue + ((size_t)rwg * 4 + r) * QREGION;
            // back: the entries were appended from the far end of the region (k_raster_q<OBJ>: object-box pixels)
            const uint32_t ent = have[j] ? w_queue[back ? QREGION - 1 - li : li] : 0u;
            pedge[j] = (ent & QE_PLANE_EDGE) != 0u || !back;
            const int lp = ent & 255;
            pix[j] = (tile_y0 + r * (WAVE_PIX / WAVE_W) + lp / WAVE_W) * R.W + tile_x0 + lp % WAVE_W;   // entries only exist for in-image pixels
            if (!have[j]) pix[j] = 0;
            const float4 l = reinterpret_cast<const float4*>(R.lut)[pix[j]];
            nxv[j] = l.x; nyv[j] = l.y;
            const float pcx = (l.x + 1.f) * 0.5f * (float)R.W, pcy = (1.f - l.y) * 0.5f * (float)R.H;
            if (have[j]) { x0 = fminf(x0, pcx); x1 = fmaxf(x1, pcx); y0 = fminf(y0, pcy); y1 = fmaxf(y1, pcy); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { zbest[j][q] = 0.f; tbest[j][q] = -1; }   // w = 1 / depth: 0 = nothing yet
          }
          unsigned long long hm = hm0;
#ifdef DT_RO_NOSTREAM
          hm = 0ull;
#endif
          if (hm) {
            // ---- mesh pass: stream the triangles of the objects whose screen box meets the tile, 64 at a time (one per
            // lane, coverage half only), keep those whose box meets the bounding box of the unit's pixels, compact them
            // into the wavefront-local LDS chunk and z-buffer the chunk against every batch of the group.
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
              x0 = fminf(x0, __shfl_xor(x0, d)); x1 = fmaxf(x1, __shfl_xor(x1, d));
              y0 = fminf(y0, __shfl_xor(y0, d)); y1 = fmaxf(y1, __shfl_xor(y1, d));
            }
            int first = 0, count = 0, t0 = 0;
            auto advance = [&]() -> bool {             // wave-uniform: next (object, t0); false when exhausted
              t0 += 64;
              while (t0 >= count) {
                if (!hm) return false;
                const uint2 fc = rng[__builtin_ctzll(hm)];
                hm &= hm - 1ull;
                first = (int)fc.x; count = (int)fc.y; t0 = 0;
              }
              return true;
            };
            // Round 3: the cull reads the triangles' screen boxes from their own contiguous array (16 B per triangle, written by
            // k_obj_setup) and only the survivors load their 64-byte coverage record: the stream used to touch one 128-byte
            // line per triangle for every (tile, env) unit an object's box meets.
            const float4* boxes = R.tribox + (size_t)env_p * R.max_tris;
            auto fetch = [&](float4& bb, int& idx) -> bool {     // this lane's triangle of the current chunk: box + index
              const bool in = t0 + lane < count;
              idx = first + t0 + lane;
              if (in) bb = boxes[idx];
              return in;
            };
            int fill = 0;
            float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
            int cur_idx = 0, nxt_idx = 0;
            bool more = advance();
            bool cur_in = more ? fetch(cur, cur_idx) : false;
            while (more) {
              const bool has_next = advance();
              const bool nxt_in = has_next ? fetch(nxt, nxt_idx) : false;
              const bool pass = cur_in && !(cur.x > x1 || cur.y < x0 || cur.z > y1 || cur.w < y0);
              const unsigned long long pm = __ballot(pass);
              if (pass) w_tris[fill + __popcll(pm & ((1ull << lane) - 1ull))] = *reinterpret_cast<const TriCov*>(base + cur_idx);
              fill += __popcll(pm);
              if (fill > TRI_CAP - 64 || (!has_next && fill > 0)) {   // chunk full, or the last one: z-buffer it
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                  if (j > 0 && g0 + j * 64 >= n_p) break;            // wave-uniform
#ifdef DT_RO_NOZ
                  if (fill == 12345) zbest[j][0] = 0.f;
#else
                  zbuffer_chunk(w_tris, w_scr, fill, have[j], lane, (nxv[j] + 1.f) * 0.5f * (float)R.W, (1.f - nyv[j]) * 0.5f * (float)R.H,
                                zbest[j], tbest[j],
#ifdef DT_RO_STATS
                                reinterpret_cast<int32_t*>(reinterpret_cast<char*>(R.pixtab) + (size_t)R.W * R.H * 64 + 1024 + 768));
#else
                                nullptr);
#endif
#endif
                }
                fill = 0;
              }
              cur = nxt; cur_idx = nxt_idx; cur_in = nxt_in; more = has_next;
            }
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if (j > 0 && g0 + j * 64 >= n_p) break;                  // wave-uniform
            // a pixel no mesh triangle covers, and that is no plane edge either, already holds its colour (the raster's)
            have[j] = have[j] && (pedge[j] || (tbest[j][0] & tbest[j][1] & tbest[j][2] & tbest[j][3]) >= 0);   // all four < 0  <=>  the AND is negative
            if (!__ballot(have[j])) continue;                        // wave-uniform: nothing of this batch needs shading
            if (have[j]) {
#ifdef DT_RO_NOSHADE
              const uint32_t v = (uint32_t)tbest[j][0] ^ (uint32_t)tbest[j][1] ^ (uint32_t)tbest[j][2] ^ (uint32_t)tbest[j][3];
#else
              const uint32_t v = shade_msaa<true>(c, m, R, s_tiles, nxv[j], nyv[j], base, zbest[j], tbest[j]);
#endif
              uint8_t* dst = R.frames + ((size_t)env_p * npix + pix[j]) * 3;
              dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
            }
          }
        }
      }
#ifdef DT_WAVE_SPANS
      { const unsigned long long d = wall_clock64() - span_i0; span_long = d > span_long ? d : span_long; span_sum += d; span_last = d; if (span_n == 1) { span_first = d; span_first_at = span_i0 - span_t0; } }
#endif
    }
  }
#ifdef DT_WAVE_SPANS
  if (R.spans && lane == 0) {
    unsigned long long* o = R.spans + ((size_t)(2048 + blockIdx.x) * 4 + wave) * 8;
    o[0] = span_t0; o[1] = wall_clock64(); o[2] = (unsigned long long)span_n; o[3] = span_long; o[4] = span_first; o[5] = span_first_at; o[6] = span_sum; o[7] = span_last;
  }
#endif
}


// ---- k_overlay_lines: the reference's GL_LINE overlays as a post-pass on the resolved frames --------------------------------------
// draw_curve (simulator.py:1886-1904, graphics.py:336-349: the lane curves of every drivable tile, 19 segments each, the one most aligned
// with the heading red, the others blue) and draw_bbox (simulator.py:1907-1918, objects.py:131-146: the collision rectangles of the
// objects and of the agent at y = 0.01, red).  The host states the segments in WORLD space (dtsim_draw_lines); here every segment is
// taken through the env's camera (EnvCam: shared or domain-randomised), clipped at the near plane, and rasterised as GL rasterises a
// 1-pixel line under multisampling: the rectangle of width 1 px around the projected segment, coverage per MSAA sample.  The pass works on
// the RESOLVED frame, per OUTPUT pixel (its source pixel comes from the LUT, so the fisheye remap is honoured): a pixel with n of its four
// samples covered becomes ((4 - n) frame + sum of the covering lines' colours) / 4 -- what the multisample resolve gives when the line is
// the nearest thing in the pixel.  Documented deviations from the GL state machine: lines are not depth-tested against
// mesh objects (they lie 1 cm above the tile plane: in front of tiles and ground always); their colour is the glColor lit as a surface
// with normal +y (GL_COLOR_MATERIAL, the normal the tile draw leaves behind), the texel of whatever texture is still bound is not applied;
// where two lines cover one sample the first in the list wins (depth func LESS on equal depths).
struct OvSeg { float ax, ay, bx, by, inv_len2, r, g, b; };     // projected segment in source-pixel space, lit colour 0..255
__global__ __launch_bounds__(256) void k_overlay_lines(RenderParams R, const EnvCam* __restrict__ cams, const float* __restrict__ lines,
                                                       const int first, const int count, const int env) {
  __shared__ OvSeg s_seg[256];
  const int tid = threadIdx.x;
  const int pix = blockIdx.x * 256 + tid;
  const EnvCam c = cams[env];
  float sxp = 0.f, syp = 0.f;
  bool live = false;
  if (pix < R.W * R.H) {
    const float4 l = reinterpret_cast<const float4*>(R.lut)[pix];
    live = l.z != 0.f;
    sxp = (l.x + 1.f) * 0.5f * (float)R.W; syp = (1.f - l.y) * 0.5f * (float)R.H;       // centre of the source pixel
  }
  const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
  const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
  float acc[3] = {0.f, 0.f, 0.f};
  uint32_t covered = 0u;
  for (int base = 0; base < count; base += 256) {
    __syncthreads();
    {                                                  // stage up to 256 projected segments
      OvSeg sg; sg.ax = sg.ay = sg.bx = sg.by = 0.f; sg.inv_len2 = -1.f; sg.r = sg.g = sg.b = 0.f;
      if (base + tid < count) {
        const float* L = lines + (size_t)(first + base + tid) * 9;
        float pe[2][3], lit[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float rx = L[3 * v] - c.Cx, ry = L[3 * v + 1] - c.Cy, rz = L[3 * v + 2] - c.Cz;
          const float xla = rx * c.sa + rz * c.ca, zla = -(rx * c.ca - rz * c.sa);
          pe[v][0] = xla; pe[v][1] = ry * c.cth - zla * c.sth; pe[v][2] = ry * c.sth + zla * c.cth;      // eye space, -z forward
        }
        float w0 = -pe[0][2], w1 = -pe[1][2];
        const float wn = NEAR_Z * 1.0001f;
        bool ok = !(w0 < wn && w1 < wn);
        if (ok && (w0 < wn || w1 < wn)) {              // clip the end behind the near plane
          const int vb = w0 < wn ? 0 : 1, vf = 1 - vb;
          const float t = (wn - (-pe[vb][2])) / ((-pe[vf][2]) - (-pe[vb][2]));
#pragma unroll
          for (int k = 0; k < 3; ++k) pe[vb][k] += t * (pe[vf][k] - pe[vb][k]);
          w0 = -pe[0][2]; w1 = -pe[1][2];
        }
        if (ok) {
#pragma unroll
          for (int v = 0; v < 2; ++v) {                // lit as a surface with world normal +y: eye-space normal (0, cth, sth)
            float ndl;
            if (c.L[3] == 0.f) ndl = c.cth * c.L[1] + c.sth * c.L[2];
            else {
              const float lx = c.L[0] - pe[v][0], ly = c.L[1] - pe[v][1], lz = c.L[2] - pe[v][2];
              ndl = (c.cth * ly + c.sth * lz) * rsqrtf(lx * lx + ly * ly + lz * lz);
            }
            lit[v] = fmaxf(ndl, 0.f);
          }
          const float ndl = 0.5f * (lit[0] + lit[1]);  // one colour per segment: the light at its middle
          sg.ax = (pe[0][0] / w0 / c.tx + 1.f) * 0.5f * (float)R.W; sg.ay = (1.f - pe[0][1] / w0 / c.ty) * 0.5f * (float)R.H;
          sg.bx = (pe[1][0] / w1 / c.tx + 1.f) * 0.5f * (float)R.W; sg.by = (1.f - pe[1][1] / w1 / c.ty) * 0.5f * (float)R.H;
          const float dx = sg.bx - sg.ax, dy = sg.by - sg.ay, l2 = dx * dx + dy * dy;
          sg.inv_len2 = l2 > 1e-12f ? 1.f / l2 : -1.f;
          sg.r = 255.f * fminf(L[6] * (c.base[0] + c.dif[0] * ndl), 1.f);
          sg.g = 255.f * fminf(L[7] * (c.base[1] + c.dif[1] * ndl), 1.f);
          sg.b = 255.f * fminf(L[8] * (c.base[2] + c.dif[2] * ndl), 1.f);
        }
      }
      s_seg[tid] = sg;
    }
    __syncthreads();
    if (!live) continue;
    const int n = min(256, count - base);
    for (int i = 0; i < n; ++i) {
      const OvSeg sg = s_seg[i];
      if (sg.inv_len2 < 0.f) continue;
      if (sxp < fminf(sg.ax, sg.bx) - 1.f || sxp > fmaxf(sg.ax, sg.bx) + 1.f || syp < fminf(sg.ay, sg.by) - 1.f || syp > fmaxf(sg.ay, sg.by) + 1.f) continue;
      const float dx = sg.bx - sg.ax, dy = sg.by - sg.ay;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float qx = sxp + ox[q] - sg.ax, qy = syp + oy[q] - sg.ay;
        const float u = (qx * dx + qy * dy) * sg.inv_len2;                       // along the segment, 0..1
        const float cr = qx * dy - qy * dx;                                      // |cross| = distance x length
        const bool in = u >= 0.f && u <= 1.f && cr * cr * sg.inv_len2 <= 0.25f;   // within half a pixel of the line
        if (in && !((covered >> q) & 1u)) { covered |= 1u << q; acc[0] += sg.r; acc[1] += sg.g; acc[2] += sg.b; }
      }
    }
  }
  if (live && covered) {
    const float nf = (float)(4 - __popc(covered));
    uint8_t* dst = R.frames + ((size_t)env * R.W * R.H + pix) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k] = (uint8_t)(fminf(fmaxf(0.25f * (nf * (float)dst[k] + acc[k]), 0.f), 255.f) + 0.5f);
  }
}


// ---- k_overlay_leds: the additively blended LED spheres of enable_leds as a post-pass on the resolved frames --------------------------
// objects.py:68-121: after a duckiebot's mesh, WorldObj.render_mesh draws per LED a 1 cm gluSphere at alpha 1 and a halo at alpha 0.2 with
// glBlendFunc(GL_SRC_ALPHA, GL_ONE), depth test and depth writes on, lighting on, no texture.  The host states the spheres in WORLD space in
// draw order (dtsim_draw_leds: centre, radius, glColor, alpha).  Per OUTPUT pixel (source pixel through the LUT) and MSAA sample: the depth of
// the opaque scene is recomputed -- the planes by classify(), the meshes by z-buffering the env's projected triangles (ScreenTri, left by the
// render pass's k_obj_setup) -- and every sphere whose FRONT surface is nearer than the sample's depth adds alpha x its lit colour (GL_COLOR_MATERIAL:
// colour x (ambient + diffuse N.L) at the hit, clamped) and writes its depth; the pixel becomes frame + sum / 4.  Additive blending is linear, so
// adding to the resolved frame equals blending into the samples.  Documented deviations (oracle/raster.py overlay_leds states the same
// interpretation): the analytic sphere for the 10 x 10 tessellation, front surfaces only (a back face that gluSphere happens to
// emit before the front face would add a second layer), lit per sample instead of per vertex, and all spheres after ALL opaque objects (GL
// interleaves them with the objects in map order).  Not a hot path: one thread per pixel, the triangle loop only under a sphere's screen box.
struct LedS { float ex, ey, ez, r, cr, cg, cb, a, bx0, bx1, by0, by1; };     // eye-space centre, radius, colour, alpha, source-pixel box
__global__ __launch_bounds__(256) void k_overlay_leds(RenderParams R, const EnvCam* __restrict__ cams, const float* __restrict__ spheres,
                                                      const int first, const int count, const int env) {
  __shared__ LedS s_led[64];
  const int tid = threadIdx.x;
  const int pix = blockIdx.x * 256 + tid;
  const EnvCam c = cams[env];
  const MapU m = map_u(R.maps[c.map_id]);
  const TileLds* tiles = reinterpret_cast<const TileLds*>(R.tile_recs);
  float nx = 0.f, ny = 0.f, sxp = 0.f, syp = 0.f;
  bool live = false;
  if (pix < R.W * R.H) {
    const float4 l = reinterpret_cast<const float4*>(R.lut)[pix];
    live = l.z != 0.f;
    nx = l.x; ny = l.y;
    sxp = (l.x + 1.f) * 0.5f * (float)R.W; syp = (1.f - l.y) * 0.5f * (float)R.H;       // centre of the source pixel
  }
  const float ox[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
  const float oy[4] = {0.375f, 0.125f, -0.125f, -0.375f};   // GL_SAMPLE_POSITION of Mesa llvmpipe, rows flipped (+y down the image)
  const float sxn = 2.f / (float)R.W, syn = 2.f / (float)R.H;
  float add[3] = {0.f, 0.f, 0.f};
  float depth[4] = {0.f, 0.f, 0.f, 0.f};
  bool have_depth = false, any = false;
  for (int base = 0; base < count; base += 64) {
    __syncthreads();
    if (tid < 64) {
      LedS sp; sp.r = -1.f; sp.ex = sp.ey = sp.ez = sp.cr = sp.cg = sp.cb = sp.a = 0.f; sp.bx0 = sp.by0 = 1.f; sp.bx1 = sp.by1 = 0.f;
      if (base + tid < count) {
        const float* S = spheres + (size_t)(first + base + tid) * 8;
        const float rx = S[0] - c.Cx, ry = S[1] - c.Cy, rz = S[2] - c.Cz;
        const float xla = rx * c.sa + rz * c.ca, zla = -(rx * c.ca - rz * c.sa);
        sp.ex = xla; sp.ey = ry * c.cth - zla * c.sth; sp.ez = ry * c.sth + zla * c.cth;                 // eye space, -z forward
        const float w = -sp.ez, r = S[3];
        if (r > 0.f && w + r > NEAR_Z) {
          sp.r = r; sp.cr = S[4]; sp.cg = S[5]; sp.cb = S[6]; sp.a = S[7];
          const float wn = fmaxf(w - r, NEAR_Z);
          const float cx = (sp.ex / w / c.tx + 1.f) * 0.5f * (float)R.W, cy = (1.f - sp.ey / w / c.ty) * 0.5f * (float)R.H;
          const float rad = r / wn / fminf(c.tx / (float)R.W, c.ty / (float)R.H) * 0.5f * 1.5f + 2.f;     // conservative: only prunes
          sp.bx0 = cx - rad; sp.bx1 = cx + rad; sp.by0 = cy - rad; sp.by1 = cy + rad;
          if (!(w > 0.f)) { sp.bx0 = sp.by0 = -1e30f; sp.bx1 = sp.by1 = 1e30f; }                          // centre behind the eye: no box, test everything
        }
      }
      s_led[tid] = sp;
    }
    __syncthreads();
    if (!live) continue;
    const int n = min(64, count - base);
    for (int i = 0; i < n; ++i) {
      const LedS sp = s_led[i];
      if (!(sp.r > 0.f) || sxp < sp.bx0 || sxp > sp.bx1 || syp < sp.by0 || syp > sp.by1) continue;
      if (!have_depth) {                               // the opaque scene's depth at the four samples, once per pixel that any sphere may touch
        have_depth = true;
        float wbest[4] = {0.f, 0.f, 0.f, 0.f};
        int tbest[4] = {-1, -1, -1, -1};
        if (R.stris && R.objbox) {
          const ObjBox* boxes = R.objbox + (size_t)env * DTSIM_MAX_OBJECTS;
          const ScreenTri* tb = R.stris + (size_t)env * R.max_tris;
          const int n_obj = R.maps[c.map_id].n_obj;
          for (int o = 0; o < n_obj; ++o) {
            const ObjBox ob = boxes[o];
            if (ob.count <= 0 || sxp < ob.bx0 - 1.f || sxp > ob.bx1 + 1.f || syp < ob.by0 - 1.f || syp > ob.by1 + 1.f) continue;
            for (int t = ob.first; t < ob.first + ob.count; ++t) test_tri(tb[t], sxp, syp, wbest, tbest);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const Ray rs = make_ray(nx + ox[q] * sxn, ny - oy[q] * syn, c.tx, c.ty, c.sth, c.cth);
          const Hit h = classify(c, m, tiles, rs);
          float d = h.cls == CLS_SKY ? 3.0e38f : h.t;
          if (tbest[q] >= 0) d = fminf(d, 1.f / wbest[q]);
          depth[q] = d;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dx = (nx + ox[q] * sxn) * c.tx, dy = (ny - oy[q] * syn) * c.ty;       // eye-space ray (dx, dy, -1) t, t = eye depth
        const float a = dx * dx + dy * dy + 1.f;
        const float b = dx * sp.ex + dy * sp.ey - sp.ez;
        const float cc = sp.ex * sp.ex + sp.ey * sp.ey + sp.ez * sp.ez - sp.r * sp.r;
        const float disc = b * b - a * cc;
        if (!(disc > 0.f)) continue;
        const float t = (b - sqrtf(disc)) / a;
        if (!(t >= NEAR_Z && t <= FAR_Z && t < depth[q])) continue;
        const float px = t * dx, py = t * dy, pz = -t;
        const float inv_r = 1.f / sp.r;
        const float nxe = (px - sp.ex) * inv_r, nye = (py - sp.ey) * inv_r, nze = (pz - sp.ez) * inv_r;
        float ndl;
        if (c.L[3] == 0.f) ndl = nxe * c.L[0] + nye * c.L[1] + nze * c.L[2];
        else {
          const float lx = c.L[0] - px, ly = c.L[1] - py, lz = c.L[2] - pz;
          ndl = (nxe * lx + nye * ly + nze * lz) * rsqrtf(lx * lx + ly * ly + lz * lz);
        }
        ndl = fmaxf(ndl, 0.f);
        add[0] += sp.a * 255.f * fminf(sp.cr * (c.base[0] + c.dif[0] * ndl), 1.f);
        add[1] += sp.a * 255.f * fminf(sp.cg * (c.base[1] + c.dif[1] * ndl), 1.f);
        add[2] += sp.a * 255.f * fminf(sp.cb * (c.base[2] + c.dif[2] * ndl), 1.f);
        depth[q] = t;
        any = true;
      }
    }
  }
  if (live && any) {
    uint8_t* dst = R.frames + ((size_t)env * R.W * R.H + pix) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k] = (uint8_t)(fminf(fmaxf((float)dst[k] + 0.25f * add[k], 0.f), 255.f) + 0.5f);
  }
}

}  // namespace

void dt_launch_overlay_leds(hipStream_t s, const RenderParams& R, const float* d_spheres, int first, int count, int env) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_overlay_leds, dim3((unsigned)((R.W * R.H + 255) / 256)), dim3(256), 0, s, R, reinterpret_cast<const EnvCam*>(R.envcam), d_spheres, first, count, env);
}

void dt_launch_overlay_lines(hipStream_t s, const RenderParams& R, const float* d_lines, int first, int count, int env) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_overlay_lines, dim3((unsigned)((R.W * R.H + 255) / 256)), dim3(256), 0, s, R, reinterpret_cast<const EnvCam*>(R.envcam), d_lines, first, count, env);
}

// Workgroups of `kernel` (RB threads, lds bytes of dynamic LDS) the device holds at once: the launch size of the persistent exact-path kernels.
template <class K> static size_t resident_blocks(K kernel, size_t lds) {
  static std::mutex mu;
  static std::map<std::tuple<int, size_t, const void*>, size_t> cache;   // (the template is instantiated per function-pointer TYPE: two kernels of one signature share it)
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_tuple(dev, lds, reinterpret_cast<const void*>(kernel));
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, n_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RB, lds) != hipSuccess || per_cu < 1) per_cu = 3;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
  if (getenv("DTSIM_DEBUG_RESIDENT")) fprintf(stderr, "[dtsim] resident_blocks: %d workgroups per CU x %d CUs (%zu B of dynamic LDS)\n", per_cu, n_cu, lds);
  return cache[key] = (size_t)per_cu * (size_t)n_cu;
}

// One range of chunks through its raster (stream s) and exact-path kernels (stream s_res, after event ev when it is
// another stream): the whole batch, or one of dt_launch_render's render parts (every array already moved to the range).
static void launch_raster_resolve(hipStream_t s, hipStream_t s_res, hipEvent_t ev, const RenderParams& R, EnvCam* cams, EnvFast* fasts, EnvQ* envq,
                                  EnvV* envv, EnvD* envd, uint8_t* frames_raster, bool quad, bool v3, bool v3dr, bool obj, bool has_pos) {
  const int n_chunks = (R.N + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
  const size_t lds = (size_t)R.n_tile_recs * sizeof(TileLds);
  const size_t lds1 = lds + (size_t)RB * PPT * sizeof(uint32_t);          // + store transpose
  const size_t lds2 = lds + (size_t)(RB / 64) * ENVS_PER_BLOCK * sizeof(EnvCam);
  const dim3 grid((unsigned)(dt_raster_tiles(R.W, R.H) * n_chunks));
  // XCD-affine map: 8 slices of ceil(n_chunks / 8) chunks, frame tiles in groups of dt_q_tile_group() (the last group padded)
  const int q_tg = dt_q_tile_group((int)dt_raster_tiles(R.W, R.H));
  const dim3 gridq((unsigned)(((dt_raster_tiles(R.W, R.H) + q_tg - 1) / q_tg) * q_tg * ((n_chunks + 7) / 8) * 8));
#define LAUNCH_RASTER(DR_, OBJ_)                                                                              \
  hipLaunchKernelGGL((k_raster<DR_, OBJ_>), grid, dim3(RB), lds1, s, R, cams, fasts, frames_raster, R.texels,               \
                     reinterpret_cast<const float4*>(R.lut), R.maps, R.tile_recs, R.queue, R.qcount)
  if (quad) {
    const size_t ldsq = (size_t)R.n_qtiles * 8 + (size_t)RB * PPT * sizeof(uint32_t);
    PixTab* pixtab = reinterpret_cast<PixTab*>(R.pixtab);
    SampTab* samptab = reinterpret_cast<SampTab*>(pixtab + (size_t)R.W * R.H);
#define LAUNCH_Q(OBJ_, S256_) hipLaunchKernelGGL((k_raster_q<OBJ_, S256_>), gridq, dim3(RB), ldsq, s, R, cams, fasts, envq, frames_raster, R.qtex, \
                                           reinterpret_cast<const float4*>(R.lut), pixtab, samptab, R.qtiles, R.queue, R.qcount)
    const bool s256 = R.qlog2 == 8 && R.qmax_tiles < 256;
    // k_raster_v3 (render_v3.inc): S = 256 textures, padded grids up to 32 x 24 tiles, up to 4 maps (else k_raster_q)
    if (v3) {
      const size_t lds3 = (size_t)R.q3_rows * V3_TAB_PITCH * 4 + (size_t)(RB / 64) * V3_WAVE_LDS * 4 + (size_t)ENVS_PER_BLOCK * sizeof(EnvQ);   // tile table, per-wavefront buffers, the chunk's EnvQ records
#define LAUNCH_V3(OBJ_) hipLaunchKernelGGL((k_raster_v3<OBJ_>), gridq, dim3(RB), lds3, s, R, cams, fasts, envq, envv, frames_raster, R.qtex, \
                                           reinterpret_cast<const float4*>(R.lut), pixtab, samptab, R.qtiles, R.queue, R.qcount)
      if (obj) LAUNCH_V3(true); else
      LAUNCH_V3(false);
#undef LAUNCH_V3
    } else if (obj) { if (s256) LAUNCH_Q(true, true); else LAUNCH_Q(true, false); }
    else { if (s256) LAUNCH_Q(false, true); else LAUNCH_Q(false, false); }
#undef LAUNCH_Q
  } else if (v3dr) {
    const size_t ldsd = (size_t)R.q3_rows * V3_TAB_PITCH * 4 + (size_t)(RB / 64) * RQ_LIST * 4;
    if (obj) hipLaunchKernelGGL((k_raster_v3dr<true>), gridq, dim3(RB), ldsd, s, R, cams, envd, frames_raster, R.qtex, reinterpret_cast<const float4*>(R.lut), R.qtiles, R.queue, R.qcount);
    else hipLaunchKernelGGL((k_raster_v3dr<false>), gridq, dim3(RB), ldsd, s, R, cams, envd, frames_raster, R.qtex, reinterpret_cast<const float4*>(R.lut), R.qtiles, R.queue, R.qcount);
  } else if (R.domain_rand || R.segment) { if (obj) LAUNCH_RASTER(true, true); else LAUNCH_RASTER(true, false); }   // per-env EnvCam path
  else { if (obj) LAUNCH_RASTER(false, true); else LAUNCH_RASTER(false, false); }
#undef LAUNCH_RASTER
  // exact path.  Quad pipeline: the plane-edge pixels were resolved inside k_raster_q (resolve_region); generic raster:
  // k_resolve drains them (front of the queue regions).  Pixels inside mesh-object screen boxes (far end of the regions)
  // are k_resolve_obj's, after either raster.
  if (!R.no_msaa) {
    if (s_res != s && (obj || !quad)) { (void)hipEventRecord(ev, s); (void)hipStreamWaitEvent(s_res, ev, 0); }
    // persistent wavefronts pulling work items: enough workgroups to fill every CU at the kernel's occupancy
    // (round 4: EXACTLY the resident workgroups -- every wavefront's first grab is static, a workgroup that waits for a slot would sit on its items)
    if (v3dr) {                                        // plane-edge pixels of k_raster_v3dr: on the quad records, through the env's homography
      const size_t ldsr = (size_t)R.q3_rows * V3_TAB_PITCH * 4;
      const dim3 rgrid((unsigned)std::min<size_t>(grid.x, resident_blocks(k_resolve_dr, ldsr)));
      hipLaunchKernelGGL(k_resolve_dr, rgrid, dim3(RB), ldsr, s_res, R, cams, envd, R.qtex, reinterpret_cast<const float4*>(R.lut), R.qtiles, R.queue, R.qcount);
    } else if (!quad) {
      const dim3 rgrid((unsigned)std::min<size_t>(grid.x, resident_blocks(k_resolve, lds2)));
      hipLaunchKernelGGL(k_resolve, rgrid, dim3(RB), lds2, s_res, R, cams, R.queue, R.qcount);
    }
    if (obj) {
      const size_t lds4 = lds + (size_t)(RB / 64) * TRI_CAP * sizeof(TriCov) + (size_t)(RB / 64) * RO_SCR_BYTES;
      const dim3 rgrid((unsigned)std::min<size_t>(grid.x, resident_blocks(k_resolve_obj<DT_RES_NB>, lds4)));
      hipLaunchKernelGGL(k_resolve_obj<DT_RES_NB>, rgrid, dim3(RB), lds4, s_res, R, cams, R.queue, 1, has_pos ? envq : (const EnvQ*)nullptr);
    }
  }
}

int dt_launch_render(hipStream_t s, const SimArrays& A, const RenderParams& R_in, int tables, const RenderOverlap* ov) {
  RenderParams R = R_in;
  tables &= 3;                                         // bit 2 (returned): this pass ran in k_env_sort's order (DTSIM_FIELD_RENDER_POS)
  EnvCam* cams = reinterpret_cast<EnvCam*>(R.envcam);
  EnvFast* fasts = reinterpret_cast<EnvFast*>(cams + A.N);
  EnvQ* envq = reinterpret_cast<EnvQ*>(fasts + A.N);
  EnvV* envv = reinterpret_cast<EnvV*>(R.envv);
  EnvD* envd = reinterpret_cast<EnvD*>(R.envd);
  // domain randomisation on the quad records (k_raster_v3dr): same table / texture conditions as k_raster_v3
  const bool v3dr = R.qtex && R.envd && R.domain_rand && !R.segment && !R.no_msaa && (R.W & 3) == 0 && R.qlog2 == 8 && R.q3_rows > 0 &&
                    R.q3_rows <= 24 && R.n_maps * 32 <= 128;
  // quad-layout fast path: shared camera, square power-of-two tile textures (else the generic k_raster)
  // (S = 256 tables carry the record-offset mask of the S256 kernels -- q8_rec256 --, which need a padded grid under 128 tiles: Q8_SNAP)
  const bool quad = R.qtex && !R.domain_rand && !R.segment && !R.no_msaa && (size_t)R.n_qtiles * 8 <= 32768 && (R.W & 3) == 0 &&
                    !(R.qlog2 == 8 && R.qmax_tiles >= 256);
  const bool obj = R.max_tris > 0;
  // render order (k_env_sort): the quad pipeline indexes by position (EnvQ, object masks, queue entries); env ids come
  // from EnvQ.env
  int32_t* pos = ((quad || v3dr) && R.envpos && A.N > ENVS_PER_BLOCK) ? R.envpos : nullptr;   // one chunk: the order does not matter
  if (pos) { hipLaunchKernelGGL(k_env_sort, dim3(1), dim3(1024), 0, s, A, R.maps, pos); tables |= 4; }
  hipLaunchKernelGGL(k_cam_setup, dim3((A.N + 63) / 64), dim3(64), 0, s, A, R.domain_rand, R.segment,
                     (float)R.W / (float)R.H, cams, fasts, R.maps, (quad || v3dr) ? envq : nullptr, R.qlog2, pos, quad ? envv : nullptr,
                     v3dr ? envd : nullptr, R.W, R.H);
  (void)hipMemsetAsync(R.work, 0, DT_WORK_INTS * sizeof(int32_t), s);            // work-item counts + cursors of k_resolve / k_resolve_obj
  if (R.max_tris > 0) {
    if (!(tables & 2)) hipLaunchKernelGGL(k_blk_setup, dim3((unsigned)dt_raster_tiles(R.W, R.H)), dim3(RB), 0, s, R, reinterpret_cast<const float4*>(R.lut), reinterpret_cast<float4*>(R.blockbox));
    tables |= 2;
    hipLaunchKernelGGL(k_obj_setup, dim3(A.N), dim3(DT_OBJSETUP_T), 0, s, A, R, cams, pos);
  }

  const int n_chunks = (R.N + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
  const bool v3 = quad && R.qlog2 == 8 && R.q3_rows > 0 && R.q3_rows <= V3_MAX_ROWS && R.n_maps * V3_MAP_COLS <= V3_TAB_PITCH / 2 &&
                  true;
  if (quad && !(tables & 1)) {
    PixTab* pixtab = reinterpret_cast<PixTab*>(R.pixtab);
    SampTab* samptab = reinterpret_cast<SampTab*>(pixtab + (size_t)R.W * R.H);
    hipLaunchKernelGGL(k_pix_setup, dim3((R.W * R.H + 255) / 256), dim3(256), 0, s, R, reinterpret_cast<const float4*>(R.lut), pixtab, samptab);
    tables |= 1;
  }
  // Render parts (DTSIM_RENDER_PARTS = P > 1): the chunks of the batch in P ranges; the raster of range p + 1 on the
  // caller's stream runs beside the exact-path kernels of range p on a second stream (they wait on memory, the raster on
  // the vector ALU and the L1).  Every per-position array is addressed relative to the range's first chunk, so the
  // kernels are the same; only the quad-record paths in the sorted render order are split (k_raster_v3, k_raster_v3dr).
  int parts = 1;
  if (ov && ov->parts > 1 && !R.no_msaa && (v3 || v3dr) && pos && (obj || !quad)) parts = std::min(std::min(ov->parts, DT_MAX_RENDER_PARTS), n_chunks / 8);
  if (parts <= 1) { launch_raster_resolve(s, s, nullptr, R, cams, fasts, envq, envv, envd, R.frames, quad, v3, v3dr, obj, pos != nullptr); return tables; }
  static const bool parts_serial = [] { const char* v = getenv("DTSIM_RENDER_PARTS_SERIAL"); return v && v[0] == '1'; }();   // experiment: the split without the overlap
  const size_t n_tiles = dt_raster_tiles(R.W, R.H), n_blk = n_tiles * 4;
  (void)hipMemsetAsync(R.work, 0, DT_WORK_INTS * parts * sizeof(int32_t), s);
  for (int p = 0; p < parts; ++p) {
    const int c0 = (int)((long long)n_chunks * p / parts), c1 = (int)((long long)n_chunks * (p + 1) / parts);
    const size_t e0 = (size_t)c0 * ENVS_PER_BLOCK, wg0 = (size_t)c0 * n_tiles;
    RenderParams Rp = R;
    Rp.N = std::min(R.N, c1 * ENVS_PER_BLOCK) - (int)e0;
    Rp.work = R.work + DT_WORK_INTS * p;
    if (R.objmask) Rp.objmask = R.objmask + e0 * n_blk;
    Rp.queue = R.queue + wg0 * (RB / 64) * QREGION;
    Rp.qcount = R.qcount + wg0 * (RB / 64);
    if (R.qend) Rp.qend = R.qend + wg0 * (RB / 64) * ENVS_PER_BLOCK;
    Rp.items = R.items + wg0 * ITEMS_PER_WG;
    Rp.items2 = R.items2 + wg0 * ENVS_PER_BLOCK;
    // per-POSITION arrays move to the range (EnvQ / EnvV / EnvD in render order, masks, queues, items above); per-ENV arrays (EnvCam, frames,
    // screen triangles, object boxes) stay whole: the kernels reach them through the env id of the position's record
    EnvQ* envq_p = envq + e0; EnvV* envv_p = envv ? envv + e0 : nullptr; EnvD* envd_p = envd ? envd + e0 : nullptr;
    launch_raster_resolve(s, parts_serial ? s : ov->s2, ov->ev[p], Rp, cams, fasts, envq_p, envv_p, envd_p, R.frames, quad, v3, v3dr, obj, true);
  }
  (void)hipEventRecord(ov->ev[DT_MAX_RENDER_PARTS], ov->s2);
  (void)hipStreamWaitEvent(s, ov->ev[DT_MAX_RENDER_PARTS], 0);
  return tables;
}
