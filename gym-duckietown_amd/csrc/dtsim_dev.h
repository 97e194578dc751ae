// Internal device-side layout of libdtsim (not part of the C-ABI).
//
// HBM layout (DESIGN.md "Data layout"):
//   * per-env state is struct-of-arrays, one array per scalar, index = env, all carved
//     from ONE slab so a checkpoint is a single memcpy (DTSIM_FIELD_STATE_BLOB);
//     thread e of the step kernel touches element e of each array => every load/store
//     of a wavefront is one fully coalesced 512-byte (f64) transaction.
//   * per-map tables are one packed blob of 8-byte words per map, copied into LDS by
//     each workgroup of the step kernel (tiles, Bezier control points, static OBBs).
//   * frames are [N][H][W][3] uint8.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dtsim.h"

// ---- constants of the reference (simulator.py:99-177), evaluated exactly as
// Python evaluates them (IEEE double, same operation order) -------------------
#define DT_CAMERA_FORWARD_DIST 0.066
#define DT_ROBOT_WIDTH (0.13 + 0.02)
#define DT_ROBOT_LENGTH 0.18
#define DT_CENTER_SHIFT (DT_CAMERA_FORWARD_DIST - (DT_ROBOT_LENGTH / 2))
#define DT_SAFETY_RAD_MULT 1.8
#define DT_AGENT_SAFETY_RAD ((DT_ROBOT_LENGTH / 2) * DT_SAFETY_RAD_MULT) /* max(L,W)=L */
#define DT_REWARD_INVALID_POSE (-1000.0)

// ---- packed map blob --------------------------------------------------------
struct MapHdr {            // 8-byte words; offsets are in words from the blob start
  int32_t grid_w, grid_h;
  int32_t n_curves, n_static;
  int32_t n_dyn, n_obj;
  int32_t off_tiles, off_curves;   // tiles: 1 word each; curves: 8 words each (P0x,P0z..P3x,P3z)
  int32_t off_heads, off_static;   // heads: 2 words per curve; static: 15 words each
  int32_t off_objs, total_words;   // objs: OBJ_WORDS words each
  int32_t n_lights, pad_l;         // traffic lights among the objects (0: k_step skips the light clock)
  double tile_size;
};
static_assert(sizeof(MapHdr) % 8 == 0, "MapHdr must be whole words");
#define MAPHDR_WORDS (sizeof(MapHdr) / 8)

struct TileRec {           // one 8-byte word
  uint8_t kind, angle, drivable, curve_cnt;
  int16_t curve_off, tex;
};
static_assert(sizeof(TileRec) == 8, "TileRec is one word");

// static collidable record: corners[8] norms[4] center[2] radius[1]
#define STATIC_WORDS 15
#define OBJ_WORDS 6      // x, z, spawn_clear, dyn_slot (-1 static, -2 optional static), light_freq, light_pattern0

struct DynInit {           // per map, per dynamic slot: initial DuckieObj state
  double cx, cz, corners[8], norm[4], heading_x, heading_z, angle, safety_radius;
  double walk_distance, vel, wait_time, wiggle;   // DuckieObj; DuckiebotObj: follow_dist, velocity, gain, trim
  int32_t obj_index, kind;                        // kind: 1 DuckieObj, 2 DuckiebotObj, 3 CheckerboardObj
};

// ---- per-env SoA ------------------------------------------------------------
struct SimArrays {
  int32_t N;
  // pose + dynamics (duckietown_world state, restated)
  double *pos_x, *pos_z, *angle;
  double *q_x, *q_y, *q_c, *q_s, *vel_u, *vel_w;
  double *ring;            // [DTSIM_MAX_DELAY][2][N]  delayed [left,right] duty
  double *war, *wal;       // angular input gains (trim)
  double *wheel_dist;
  double *timestamp, *speed;
  double *reward;
  double *lane;            // [4][N]
  double *prox;
  double *wheels;          // [2][N]
  // dynamic objects [slot][N]
  double *ob_cx, *ob_cz, *ob_sx, *ob_sz;
  double *ob_corners;      // [8][DTSIM_MAX_DYNAMIC][N]
  double *ob_vel, *ob_wait, *ob_time, *ob_angle, *ob_wiggle, *ob_yrot;
  // render / DR parameters (f32, consumed by the raster)
  float *cam;              // [6][N] height, pitch(rad), fov_y(rad), noise x,y,z
  float *colors;           // [16][N] horizon rgb, ground rgb, ambient rgb, diffuse rgb, light xyzw
  int32_t *ring_head, *step_count, *tile_i, *tile_j, *map_id, *episode;
  uint8_t *done, *done_code, *in_lane;
  uint8_t *ob_active;      // [DTSIM_MAX_DYNAMIC][N]
  uint8_t *ob_visible;     // [DTSIM_MAX_OBJECTS][N]
  uint8_t *ob_light;       // [DTSIM_MAX_OBJECTS][N] TrafficLightObj.pattern
  double *ob_cy;           // [DTSIM_MAX_DYNAMIC][N] centre height (CheckerboardObj; 0 for the others)
  double *ob_ext;          // [5][DTSIM_MAX_DYNAMIC][N] DuckiebotObj follow_dist, radius, wheel_dist, robot_width, robot_length
  double *tl_time;         // [N] TrafficLightObj.time (object clock: not reset with the env, objects.py:441,459)
};

struct StepParams {
  int32_t n_steps, frame_skip, max_steps, delay_steps;
  int32_t action_mode, actions_f64, auto_reset, n_pool;
  uint32_t step_flags;                  // DTSIM_STEP_*
  int32_t light_capture, domain_rand;   // DTSIM_F_LIGHT_CAPTURE (device-side resets take the new light through the last frame's camera); DTSIM_F_DOMAIN_RAND (that camera carries its noise)
  int32_t lanes;                        // lanes of a wavefront that share one env in k_step (1, 2, 4, 8; physics.hip Coop)
  const dtsim_reset_sampler* sampler;   // device copy, or null: device-side reset sampling (N2)
  double delta_time, robot_speed;
  double gain, trim, radius, k, limit;
};

// All maps, device side
struct MapSet {
  int32_t n_maps;
  int32_t blob_off[DTSIM_MAX_MAPS];   // word offset of each map blob inside `blobs`
  int32_t total_words;
  const uint64_t* blobs;
  const DynInit* dyn;                 // [n_maps][DTSIM_MAX_DYNAMIC]
};

// launchers implemented in physics.hip
void dt_launch_step(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P,
                    const void* actions, const dtsim_init_state* pool);
void dt_launch_reset(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P,
                     const uint8_t* mask, const dtsim_init_state* states);
void dt_launch_query(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P, int n,
                     const int32_t* env_idx, const double* poses, double safety_factor, dtsim_probe* out);

// ---- raster -----------------------------------------------------------------
struct TexDev { int32_t w, h, off, pad; };   // off: texel offset into the texel pool; storage is (h+1) x (w+1), padded for REPEAT
struct MeshDev { int32_t n_tris, off; float mn[3], mx[3]; };   // off: triangle offset into the pool; model-space AABB
struct TriDev { float v[3][3]; float n[3][3]; float c[3][3]; float uv[3][2]; int32_t tex, pad; };   // tex: texture index or -1

struct RenderMapDev {       // per map, raster view of the grid + objects
  int32_t grid_w, grid_h, n_obj, n_tris;   // n_tris: total mesh triangles of the map's objects
  float tile_size, inv_tile_size;
  int32_t tile_off;         // offset into tile table (uint32 per tile: tex | angle<<8 | present<<15)
  int32_t obj_off;          // offset into object-instance table
  int32_t qt_off, qt_pitch; // quad-texture tile table of the map: first entry, row pitch (grid_w + 2*DT_QRING)
};

struct ObjInstDev {         // static render instance (dynamic ones are patched per env)
  float x, y, z, scale, yrot_deg;
  int32_t mesh_id, dyn_slot;
  int32_t light_tris, light_tex0, light_tex1;   // traffic light: first `light_tris` triangles take texture 0 / 1 by pattern
  int32_t pad[2];
};
static_assert(sizeof(ObjInstDev) == 48, "ObjInstDev is 48 bytes");

// LDS-staged raster tile record: texel base of the (padded) texture, flags (bit0 present,
// bit1 textured), and the affine map tile-fraction (fx, fz) -> texel coordinates
// x = mxx*fx + mxz*fz + ox, y = myx*fx + myz*fz + oy encoding glRotatef(angle*90+180)
// about y, uv = (pu, 1-pv) (simulator.py:394-401,1872-1873) and the GL_LINEAR half-texel shift.
struct alignas(16) TileLds { uint32_t tex_off, flags; float mxx, mxz, ox, myx, myz, oy; };
static_assert(sizeof(TileLds) == 32, "TileLds is 32 bytes");
#define DTSIM_LDS_TILES 1024   // raster tile records of all maps together (32 KB of LDS)

// One mesh triangle of one env after model/view/projection and per-vertex lighting
// (objects.py:123-148, objmesh.py:360-375): rectilinear pixel coordinates, 1/w, lit colour/w.
struct alignas(16) ScreenTri {
  // first 64 bytes: coverage / depth (all k_resolve's z-buffer pass reads; same layout as render.hip TriCov)
  float bx0, bx1, by0, by1;  // pixel bounding box (+-1 px); empty (bx0 > bx1) when culled
  float sx[3], sy[3], iw[3];
  float inv_area;            // 0 => culled (behind the near plane / degenerate / invisible)
  int32_t index;             // position in the env's triangle order (z-buffer tie break)
  int32_t pad;
  // second 64 bytes: shading attributes, read only for the winning triangle of a sample
  float cw[3][3];            // per-vertex lit colour (0..255) divided by w
  float uw[3], vw[3];        // per-vertex texture coordinates divided by w
  int32_t tex;               // texture index of the material chunk, -1 = untextured
};
static_assert(sizeof(ScreenTri) == 128, "ScreenTri is 128 bytes");
struct ObjBox { float bx0, bx1, by0, by1; int32_t first, count, pad[2]; };          // screen box + triangle range of one object

// raster work decomposition: a workgroup owns a DT_TILE_W x DT_TILE_H pixel tile for DT_ENVS_PER_BLOCK consecutive envs (positions of the render order)
#ifndef DT_WAVE_W
#define DT_WAVE_W 128                      // pixel columns of a wavefront's block (128 x 2 pixels; 64 x 4 measured 2.5 % slower)
#endif
#ifndef DT_PPT
#define DT_PPT 4                           // pixels per lane: a wavefront's block is 64 * DT_PPT pixels
#endif
#ifndef DT_V3_WW
#define DT_V3_WW 128                       // k_raster_v3 (render_v3.inc): pixel columns of ITS wavefront block (128 / 64 / 32); the workgroup tile stays 128 x 8
#endif
#define DT_TILE_W DT_WAVE_W
#define DT_TILE_H (4 * (64 * DT_PPT / DT_WAVE_W))  // 4 wavefronts stacked vertically
#ifndef DT_ENVS_PER_BLOCK
#define DT_ENVS_PER_BLOCK 64                 // envs a raster workgroup loops over (round 6: 32 -> 64 -- the tile tables and per-pixel constants of a workgroup serve twice the envs:
                                             // C3 - 1.9 %, C5 - 2.8 %, C4 +- 0, frames unchanged; a queue entry's env field has six bits: 64 is the limit)
#endif
#ifndef DT_ITEM_B
#define DT_ITEM_B 8                          // 64-entry edge batches per k_resolve work item
#endif
#define DT_ITEMS_PER_WG (4 * (DT_PPT * DT_ENVS_PER_BLOCK) / DT_ITEM_B) // worst case: 4 regions x (64*PPT px x envs / 64) batches
// Quad-layout tile textures for the one-ray fast path (render.hip k_raster_q): per (texture, tile angle) pair one
// block of S x S records of 16 bytes, record (x0, z0) = the four GL_LINEAR taps of the pre-rotated tile texture around
// quad cell (x0, z0) as channel-planar bytes {R00 R10 R01 R11}, {G..}, {B..} + a meta dword (see DT_QMETA_*).
#define DT_QRING 4                           // ring of off-grid cells around each map's tile table, in tiles
// The pool starts with two single records every cell of a non-textured tile maps to: record 0 = off the grid (ground
// quad / sky), record 1 = present but untextured tile (exact path).
// DT_QMETA -- meta dword: low 16 bits = cells to the nearest tile boundary if the cell belongs to a textured tile (else 0),
// high 16 bits = 1 if the cell is off the grid (else 0); 0 / 0 = always the exact path.
static inline size_t dt_raster_tiles(int W, int H) {
  return (size_t)((W + DT_TILE_W - 1) / DT_TILE_W) * (size_t)((H + DT_TILE_H - 1) / DT_TILE_H);
}

struct RenderParams {
  int32_t N, W, H, distortion;
  int32_t domain_rand, n_maps, n_tile_recs, no_msaa;   // no_msaa: profiling ablation only (-DDT_RASTER_NO_MSAA build variant)
  int32_t tex_w, tex_h;           // all tile textures share one (power-of-two) size
  const TileLds* tile_recs;       // [n_tile_recs], maps concatenated (RenderMapDev.tile_off)
  uint8_t* frames;
  const float* lut;             // [H*W][4]: source-pixel NDC x, y (of the rectilinear pixel), valid flag, pad
  const uint32_t* texels;       // RGBA8 pool
  const TexDev* tex;
  const RenderMapDev* maps;
  const uint32_t* tiles;
  const ObjInstDev* objs;
  const MeshDev* meshes;
  const TriDev* tris;
  void* envcam;                 // [N] EnvCam scratch written by the setup kernel
  // mesh objects: per-env screen-space triangles written by the object setup kernel
  int32_t max_tris, segment;    // triangle slots per env (max over maps), 0 = no objects anywhere; segment: DTSIM_RENDER_SEGMENT
  ScreenTri* stris;             // [N][max_tris]
  float4* tribox;               // [N][max_tris] screen boxes (bx0, bx1, by0, by1) of the triangles: k_resolve_obj's cull stream
  ObjBox* objbox;               // [N][DTSIM_MAX_OBJECTS]
  float* blockbox;              // [raster tiles * 4][4] source-pixel bounding box of each raster wavefront block (k_blk_setup)
  unsigned long long* objmask;  // [N][raster tiles * 4] objects whose screen box meets the block (bit o), written by k_obj_setup
  uint2* objrange;              // [DTSIM_MAX_MAPS][DTSIM_MAX_OBJECTS] (first, count) of each object's triangles in its map's order (k_blk_setup)
  uint16_t* queue;              // MSAA edge-pixel queue regions, [workgroups][4][256*16]
  int32_t* qcount;              // [workgroups][4]
  uint16_t* qend;               // [workgroups][4][DT_ENVS_PER_BLOCK] queue fill of each region after each env of the chunk (mesh-object renders)
  int32_t* dbg;                 // optional debug counters (DTSIM_DEBUG_QUEUE), else null
  int32_t* work;                // [0] number of work items (raster appends), [1] resolve cursor, [2], [3] the same for k_resolve_obj ([2] = its heavy items, front of the list; [6] = the others, back); zeroed per render (DT_WORK_INTS per render part)
  uint32_t* items;              // [workgroups * DT_ITEMS_PER_WG] work items: raster workgroup * DT_ITEMS_PER_WG + part
  uint32_t* items2;             // [workgroups * DT_ENVS_PER_BLOCK] work items of k_resolve_obj: raster workgroup * DT_ITEMS_PER_WG + env group
  const uint8_t* mesh_seg;      // [n_meshes][4] flat segmentation colour per mesh (segment renders only)
  // quad-layout fast path (null qtex: the generic k_raster is used)
  const uint8_t* qtex;          // quad blocks, 16 B records
  const uint32_t* qtiles;       // [n_qtiles][2] per padded-table cell: byte offset of its block, mask of the record's offset inside it (cell mask for sizes other than 256); maps concatenated
  int32_t n_qtiles, qlog2;      // qlog2: log2(S), S = tile texture size
  float q_per_m;                // quad cells per metre (S / tile_size), max over maps: scales the MSAA margin
  int32_t qmax_tiles;           // largest padded grid extent over the maps (tiles)
  int32_t* envpos;              // [N] position of each env in the render order (k_env_sort)
  void* dump;                   // 1 KB scratch: masked lanes of the unconditional frame store write here
  void* pixtab;                 // [H*W] PixTab (16 B) then [H*W] SampTab (48 B): per-pixel tables of the shared camera
  void* envv;                   // [N + 1] EnvV (render.hip): k_raster_v3's per-env constants in render order
  void* envd;                   // [N] EnvD (render_v3dr.inc, 320 B, render order): k_raster_v3dr's per-env constants (domain randomisation)
  int32_t q3_rows;              // k_raster_v3 (render_v3.inc): rows of its LDS tile table (largest padded grid height); 0: k_raster_q is used
  int32_t pad4_;
  unsigned long long* spans;    // DT_WAVE_SPANS build variant only (else null): [2][2048 workgroups][4 wavefronts]{start, end, items, longest / first item, start of the first, sum, last item} in 100 MHz ticks
};
// tables: bit 0 = the per-pixel tables (k_pix_setup), bit 1 = block boxes / object ranges (k_blk_setup) are valid from an
// earlier launch (they depend on the camera LUT and the maps only); returns the bits that are valid after this launch,
// plus bit 2 when the pass ran in k_env_sort's render order (RenderParams.envpos holds it: DTSIM_FIELD_RENDER_POS).
// Render parts (round 4): with parts > 1 the exact-path kernels of one range of chunks run on s2 beside the raster of the
// next range; ev[p] orders range p across the two streams, ev[DT_MAX_RENDER_PARTS] joins s2 back into the caller's stream.
#define DT_MAX_RENDER_PARTS 8
#define DT_WORK_INTS 8           // RenderParams.work: ints per render part
struct RenderOverlap { int parts; hipStream_t s2; hipEvent_t ev[DT_MAX_RENDER_PARTS + 1]; };
int dt_launch_render(hipStream_t s, const SimArrays& A, const RenderParams& R, int tables, const RenderOverlap* ov = nullptr);
// GL_LINE overlays (draw_curve / draw_bbox) as a post-pass on the resolved frame of `env`: d_lines = [..][9] world-space segments + colour
// (device memory), `count` of them from `first` on; uses the EnvCam the last render wrote.
void dt_launch_overlay_lines(hipStream_t s, const RenderParams& R, const float* d_lines, int first, int count, int env);
// the LED spheres of enable_leds (render.hip k_overlay_leds): [count] spheres of env `env` from d_spheres[first..], world space, R = the last render pass's parameters
void dt_launch_overlay_leds(hipStream_t s, const RenderParams& R, const float* d_spheres, int first, int count, int env);

#ifndef DT_OBS_STAGE_ROWS
#define DT_OBS_STAGE_ROWS 8
#endif
#ifndef DT_OBS_LDS_KB
#define DT_OBS_LDS_KB 48
#endif
#ifndef DT_OBS_MAX_RPB
#define DT_OBS_MAX_RPB 8
#endif
// observation post-processing (observe.hip): Pillow-exact bilinear resize + layout + normalisation
struct ObserveParams {
  int32_t N, H, W, oh, ow;
  int32_t kx, ky;               // taps per output column / row in the tables
  int32_t rows_per_block;       // output rows per workgroup
  int32_t max_rows_in;          // input rows any workgroup needs (sizes the LDS intermediate)
  int32_t chw, f32;             // layout (0: [N,h,w,3], 1: [N,3,h,w]) and dtype (0: uint8, 1: float32 / 255)
  const uint8_t* frames;        // [N,H,W,3]
  void* out;
  const int32_t* bx;            // [ow][2] first tap, tap count   (dtsim/resample.py coeffs)
  const int32_t* kkx;           // [ow][kx] 22-bit fixed-point taps
  const int32_t* by;            // [oh][2]
  const int32_t* kky;           // [oh][ky]
  // power-of-two down-scaling (640 -> 160 / 80, 480 -> 240 / 120 / 60): away from the borders every output column (row) has the
  // SAME taps, and they are small integers times a power of two (the triangle filter of scale S normalises to (1, 3, .., 2S-1,
  // 2S-1, .., 1) / 2S^2).  hfast: 0 off, else S: the 2S taps x 3 channels of a column sit in `hn` aligned dwords starting `hoff`
  // bytes from 3 S ox; hw[c][d] holds channel c's tap weights at their byte positions of dword d (zeros elsewhere): three
  // chains of v_dot4_u32_u8 filter a column.  vfast: 0 off, else S: vw[t] the 2S row weights, applied to four bytes at a time
  // in two 16-bit lanes.  hsh / vsh: the fixed-point shift that is left (22 - log2 of the common factor).
  int32_t hfast, hn, hoff, hsh;
  int32_t vfast, vsh;
  uint32_t hw[3][12];
  uint32_t vw[16];
};
size_t dt_observe_lds_bytes(const ObserveParams& P);
void dt_launch_observe(hipStream_t s, const ObserveParams& P);
// OpenCV INTER_CUBIC: bx / by = first of the four taps per output column / row (borders replicate), kkx / kky = [..][4] 11-bit taps
void dt_launch_observe_cubic(hipStream_t s, const ObserveParams& P);
