// physics.hip -- the non-render half of Simulator.step as ONE fused HIP kernel for gfx950.
//
//   k_step : one thread per env; per launch n_steps x { [DuckietownEnv kinematics] ->
//            frame_skip x update_physics -> _compute_done_reward }.
//            Reference: envs/duckietown_env.py:36-72, simulator.py:1551-1584, 1669-1705,
//            2076-2088, collision.py, graphics.py:286-333, objects.py:384-431.
//   k_reset: Simulator.reset()'s state hand-over (simulator.py:740-755).
//   k_query: the reference's geometry queries at arbitrary poses (used by the host
//            reset sampler and by the drop-in Simulator's query methods).
//
// Numerics: float64 throughout, compiled with -ffp-contract=off so that every product
// and sum is rounded separately, in the reference's operation order (numpy elementwise
// ufuncs do not fuse).  done/collision flags and tile indices are therefore bit-exact
// against the oracle except on measure-zero ties (DESIGN.md "Parity").
//
// Memory: per-env state is SoA => all wave accesses are coalesced 512-B transactions.
// Map tables (<= a few KB per map) are staged once per workgroup into LDS.
// Roofline: HBM-bound by construction: algorithmic bytes per env-step =
// read+write of the SoA state; at N=4096 the working set is L2-resident and the kernel
// is latency-bound, so n_steps are fused per launch.
#include "dtsim_dev.h"

#pragma clang fp contract(off)

#ifndef STEP_BLOCK
#define STEP_BLOCK 64
#endif

namespace {

struct MapView {
  const MapHdr* h;
  const TileRec* tiles;
  const double* curves;
  const double* heads;
  const double* stat;
  const double* objs;
};

__device__ inline MapView map_view(const uint64_t* blob) {
  MapView v;
  v.h = reinterpret_cast<const MapHdr*>(blob);
  v.tiles = reinterpret_cast<const TileRec*>(blob + v.h->off_tiles);
  v.curves = reinterpret_cast<const double*>(blob + v.h->off_curves);
  v.heads = reinterpret_cast<const double*>(blob + v.h->off_heads);
  v.stat = reinterpret_cast<const double*>(blob + v.h->off_static);
  v.objs = reinterpret_cast<const double*>(blob + v.h->off_objs);
  return v;
}

// simulator.py:1134-1149 get_grid_coords + :1053-1063 _get_tile
__device__ inline const TileRec* tile_at(const MapView& m, double x, double z, int& i, int& j) {
  const double ts = m.h->tile_size;
  const double fi = floor(x / ts), fj = floor(z / ts);
  // keep int conversion defined for wild values
  i = (fi < -1e9) ? -1000000000 : (fi > 1e9 ? 1000000000 : (int)fi);
  j = (fj < -1e9) ? -1000000000 : (fj > 1e9 ? 1000000000 : (int)fj);
  if (i < 0 || i >= m.h->grid_w || j < 0 || j >= m.h->grid_h) return nullptr;
  const TileRec* t = &m.tiles[j * m.h->grid_w + i];
  return t->kind == DTSIM_TILE_EMPTY ? nullptr : t;
}

// simulator.py:1411-1428
__device__ inline bool drivable_pos(const MapView& m, double x, double z) {
  int i, j;
  const TileRec* t = tile_at(m, x, z, i, j);
  return t != nullptr && t->drivable;
}

// collision.py:50-61 (closed intervals)
__device__ inline bool overlaps(double min1, double max1, double min2, double max2) {
  return (min1 <= min2 && min2 <= max1) || (min2 <= min1 && min1 <= max2);
}

// min/max over 4 corners of n . c   (collision.py:37-47 tensor_sat_test)
__device__ inline void proj4(double nx, double nz, const double* c, double& mn, double& mx) {
  double p0 = nx * c[0] + nz * c[1];
  double p1 = nx * c[2] + nz * c[3];
  double p2 = nx * c[4] + nz * c[5];
  double p3 = nx * c[6] + nz * c[7];
  mn = fmin(fmin(p0, p1), fmin(p2, p3));
  mx = fmax(fmax(p0, p1), fmax(p2, p3));
}

// collision.py:129-186: SAT between the agent box (corners ac, axes an) and one OBB.
// ---- lane co-operation inside one env (DTSIM_STEP_LANES, k_step<SAMPLER, L>) -----------------------------------------
// With L > 1, L adjacent lanes of a wavefront work on ONE env: everything serial (dynamics, trigonometry, the Bezier
// bisection) runs redundantly on all of them -- identical inputs, identical results, identical (duplicate) stores -- and
// the loops over OBJECTS (collision.py:129-186 intersects / intersects_single_obj, simulator.py:1430-1459
// proximity_penalty2, objects.py step()) are split: lane `sub` takes objects sub, sub + L, ...  Flags are OR-ed over the
// group with a butterfly of permute moves; penalties are added IN OBJECT ORDER by every lane (one shuffle per object), so
// reward and proximity carry the reference's order of additions -- the same bits -- for every L.
struct Coop { int sub, L; };
__device__ inline bool grp_any(bool v, const Coop& c) {
  if (c.L == 1) return v;
  int x = v ? 1 : 0;
  for (int d = 1; d < c.L; d <<= 1) x |= __shfl_xor(x, d);
  return x != 0;
}

__device__ inline bool sat_pair(const double* ac, const double* an, const double* oc, const double* on) {
  double a0, a1, b0, b1;
  proj4(an[0], an[1], ac, a0, a1);
  proj4(an[0], an[1], oc, b0, b1);
  if (!overlaps(a0, a1, b0, b1)) return false;
  proj4(an[2], an[3], ac, a0, a1);
  proj4(an[2], an[3], oc, b0, b1);
  if (!overlaps(a0, a1, b0, b1)) return false;
  proj4(on[0], on[1], ac, a0, a1);
  proj4(on[0], on[1], oc, b0, b1);
  if (!overlaps(a0, a1, b0, b1)) return false;
  proj4(on[2], on[3], ac, a0, a1);
  proj4(on[2], on[3], oc, b0, b1);
  if (!overlaps(a0, a1, b0, b1)) return false;
  return true;
}

// collision.py:9-34 agent_boundbox applied to _actual_center (simulator.py:2112-2116)
__device__ inline void agent_corners(double px, double pz, double dx, double dz, double rx, double rz,
                                     double* c /*[8]*/) {
  const double tx = px + DT_CENTER_SHIFT * dx;
  const double tz = pz + DT_CENTER_SHIFT * dz;
  const double hw = 0.5 * DT_ROBOT_WIDTH, hl = 0.5 * DT_ROBOT_LENGTH;
  c[0] = (tx - hw * rx) - hl * dx;  c[1] = (tz - hw * rz) - hl * dz;
  c[2] = (tx + hw * rx) - hl * dx;  c[3] = (tz + hw * rz) - hl * dz;
  c[4] = (tx + hw * rx) + hl * dx;  c[5] = (tz + hw * rz) + hl * dz;
  c[6] = (tx - hw * rx) + hl * dx;  c[7] = (tz - hw * rz) + hl * dz;
}

// simulator.py:1473-1492 _collision.  Agent axes = (dir, right): the eigenvectors the
// reference gets from generate_norm (collision.py:99-106) for the 0.15 x 0.18 box.
__device__ inline bool collision(const MapView& m, const SimArrays& A, const DynInit* dyn, int e,
                                 const double* ac, double dx, double dz, double rx, double rz, const Coop co = Coop{0, 1}) {
  const double an[4] = {dx, dz, rx, rz};
  const int ns = m.h->n_static;
  bool hit = false;
  for (int s = co.sub; s < ns && !hit; s += co.L) {
    const double* r = m.stat + s * STATIC_WORDS;
    hit = sat_pair(ac, an, r, r + 8);
  }
  const int nd = m.h->n_dyn;
  const int N = A.N;
  for (int d = co.sub; d < nd && !hit; d += co.L) {
    double oc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) oc[k] = A.ob_corners[(size_t)(k * DTSIM_MAX_DYNAMIC + d) * N + e];
    hit = sat_pair(ac, an, oc, dyn[d].norm);
  }
  return grp_any(hit, co);                           // any object: the reference returns on the first hit, same boolean
}

// simulator.py:1494-1534 _valid_pose, including the second _actual_center inside
// get_agent_corners (box centred at cur_pos - 0.048 dir).
__device__ inline bool valid_pose(const MapView& m, const SimArrays& A, const DynInit* dyn, int e,
                                  double px, double pz, double angle, double sf, bool* coll_out, const Coop co = Coop{0, 1}) {
  const double ca = cos(angle), sa = sin(angle);
  const double dx = ca, dz = -sa, rx = sa, rz = ca;
  const double cx = px + DT_CENTER_SHIFT * dx, cz = pz + DT_CENTER_SHIFT * dz;
  const double w = sf * 0.5 * DT_ROBOT_WIDTH, l = sf * 0.5 * DT_ROBOT_LENGTH;
  const bool all_drivable = drivable_pos(m, cx, cz) && drivable_pos(m, cx - w * rx, cz - w * rz) &&
                            drivable_pos(m, cx + w * rx, cz + w * rz) && drivable_pos(m, cx + l * dx, cz + l * dz);
  double ac[8];
  agent_corners(cx, cz, dx, dz, rx, rz, ac);
  const bool coll = collision(m, A, dyn, e, ac, dx, dz, rx, rz, co);
  if (coll_out) *coll_out = coll;
  return (!coll) && all_drivable;
}

// simulator.py:1430-1459 proximity_penalty2 + collision.py:189-211 + objects.py:373-382
__device__ inline double proximity(const MapView& m, const SimArrays& A, const DynInit* dyn, int e,
                                   double px, double pz, double angle, const Coop co = Coop{0, 1}) {
  const double cx = px + DT_CENTER_SHIFT * cos(angle), cz = pz + DT_CENTER_SHIFT * (-sin(angle));
  const double r1 = DT_AGENT_SAFETY_RAD;
  double total = 0.0;
  const int ns = m.h->n_static;
  if (ns > 0) {
    // L lanes per env: lane `sub` scores objects sub, sub + L, ...; the scores of a round are then ADDED IN OBJECT ORDER by every
    // lane (one shuffle per object), so the sum has the reference's order of additions whatever L is (round-3 advisor: a butterfly
    // sum made the reward's last bit depend on the batch size)
    bool gate = false;
    double sum = 0.0;
    for (int s0 = 0; s0 < ns; s0 += co.L) {
      const int s = s0 + co.sub;
      double neg = 0.0;
      if (s < ns) {
        const double* r = m.stat + s * STATIC_WORDS;
        const double ddx = r[12] - cx, ddz = r[13] - cz;
        const double d = sqrt((ddx * ddx + 0.0) + ddz * ddz);
        const double r2 = r[14];
        const double d2 = d * d, lo = (r1 - r2) * (r1 - r2), hi = (r1 + r2) * (r1 + r2);
        gate = gate || (lo <= d2 && d2 <= hi) || (d < fabs(r1 - r2));
        const double score = (d - r1) - r2;
        if (score < 0) neg = score;
      }
      if (co.L == 1) sum += neg;
      else for (int j = 0; j < co.L && s0 + j < ns; ++j) sum += __shfl(neg, j, co.L);
    }
    if (co.L > 1) gate = grp_any(gate, co);
    total = gate ? sum : 0.0;
  }
  const int nd = m.h->n_dyn;
  const int N = A.N;
  for (int d0 = 0; d0 < nd; d0 += co.L) {
    const int d = d0 + co.sub;
    double pen = 0.0;
    if (d < nd) {
      const double ddx = cx - A.ob_cx[(size_t)d * N + e], ddy = 0.0 - A.ob_cy[(size_t)d * N + e], ddz = cz - A.ob_cz[(size_t)d * N + e];
      const double dist = sqrt((ddx * ddx + ddy * ddy) + ddz * ddz);     // |agent_pos - center| in 3-D (objects.py:373-382, 525)
      const double score = (dist - r1) - dyn[d].safety_radius;
      pen = fmin(0.0, score);
    }
    if (co.L == 1) total += pen;                       // the reference's order of additions ...
    else for (int j = 0; j < co.L && d0 + j < nd; ++j) total += __shfl(pen, j, co.L);   // ... for every L
  }
  return total;
}

struct Lane {
  bool in_lane;
  int curve_idx;
  double t, pt_x, pt_z, tan_x, tan_z, dist, dot_dir, angle_deg, angle_rad;
};

// graphics.py:286-297: Bernstein weights are exact doubles for t on the 1/512 grid.
__device__ inline void bezier_point(const double* cp, double t, double& x, double& z) {
  const double u = 1.0 - t;
  const double b0 = u * u * u, b1 = 3 * t * (u * u), b2 = 3 * (t * t) * u, b3 = t * t * t;
  x = ((b0 * cp[0] + b1 * cp[2]) + b2 * cp[4]) + b3 * cp[6];
  z = ((b0 * cp[1] + b1 * cp[3]) + b2 * cp[5]) + b3 * cp[7];
}

// simulator.py:1337-1409 closest_curve_point + get_lane_pos2
// (Not lane-cooperative, by measurement -- round 4, profiles/r04_c2_lane_pose_ab.txt: the curve argmax split over the L lanes of
// an env (2 / 6 / 12 dot products, a (value, index) butterfly with np.argmax's first-maximum tie break) and the bisection's two
// initial end points on two lanes leave the step where it was or 1 - 2 % slower on the 12-curve junction map: the bisection has
// ONE new curve point per level, the chain is serial.)
__device__ inline Lane lane_pos(const MapView& m, double px, double pz, double angle) {
  Lane L;
  L.in_lane = false; L.curve_idx = -1;
  L.t = L.pt_x = L.pt_z = L.tan_x = L.tan_z = L.dist = L.dot_dir = L.angle_deg = L.angle_rad = 0.0;
  int i, j;
  const TileRec* tl = tile_at(m, px, pz, i, j);
  if (tl == nullptr || !tl->drivable) return L;
  const double dx = cos(angle), dz = -sin(angle);
  // argmax_c  heading_c . dir  (headings pre-divided by the Frobenius norm on the host,
  // simulator.py:1355-1362); first maximum wins like np.argmax.
  int best = 0;
  double bestv = 0.0;
  for (int c = 0; c < tl->curve_cnt; ++c) {
    const double* hd = m.heads + 2 * (tl->curve_off + c);
    const double v = hd[0] * dx + hd[1] * dz;
    if (c == 0 || v > bestv) { bestv = v; best = c; }
  }
  const double* cp = m.curves + 8 * (tl->curve_off + best);
  // graphics.py:316-333 bezier_closest: 8-level endpoint-distance bisection, strict <
  // The reference evaluates both end points at every level; one of them is the end point kept from the level before
  // (same t, same arithmetic, same value), so only the moved one is evaluated here: 9 curve points instead of 16.
  double tb = 0.0, tt = 1.0;
  auto dist_at = [&](double t) -> double {
    double x, z;
    bezier_point(cp, t, x, z);
    return sqrt(((x - px) * (x - px) + 0.0) + (z - pz) * (z - pz));
  };
  double d_bot = dist_at(tb), d_top = dist_at(tt);
  for (int n = 0; n < 8; ++n) {
    const double mid = (tb + tt) * 0.5;
    const bool lower = d_bot < d_top;
    if (lower) tt = mid; else tb = mid;
    if (n < 7) {                                       // the last level's end points are not looked at again
      const double d_mid = dist_at(mid);
      if (lower) d_top = d_mid; else d_bot = d_mid;
    }
  }
  const double t = (tb + tt) * 0.5;
  double ptx, ptz;
  bezier_point(cp, t, ptx, ptz);
  // graphics.py:300-313 bezier_tangent
  const double u = 1.0 - t;
  const double c0 = 3 * (u * u), c1 = 6 * u * t, c2 = 3 * (t * t);
  double tx = (c0 * (cp[2] - cp[0]) + c1 * (cp[4] - cp[2])) + c2 * (cp[6] - cp[4]);
  double tz = (c0 * (cp[3] - cp[1]) + c1 * (cp[5] - cp[3])) + c2 * (cp[7] - cp[5]);
  const double nrm = sqrt((tx * tx + 0.0) + tz * tz);
  tx /= nrm; tz /= nrm;
  // simulator.py:1387-1409
  double dot = (dx * tx + 0.0) + dz * tz;
  dot = fmin(fmax(dot, -1.0), 1.0);
  const double rvx = -tz, rvz = tx;  // cross(tangent, up)
  const double dist = ((px - ptx) * rvx + 0.0) + (pz - ptz) * rvz;
  double arad = acos(dot);
  if (((dx * rvx + 0.0) + dz * rvz) < 0) arad *= -1;
  L.in_lane = true; L.curve_idx = best; L.t = t;
  L.pt_x = ptx; L.pt_z = ptz; L.tan_x = tx; L.tan_z = tz;
  L.dist = dist; L.dot_dir = dot; L.angle_rad = arad;
  L.angle_deg = arad * (180.0 / 3.141592653589793);  // np.rad2deg
  return L;
}

// simulator.py:1654-1667
__device__ inline double compute_reward(const Lane& L, double prox, double speed) {
  if (!L.in_lane) return 40 * prox;
  return ((1.0 * speed) * L.dot_dir + (-10 * fabs(L.dist))) + (40 * prox);
}

// ---- device-side reset sampler (dtsim_reset_sampler) ----------------------------------
// Philox4x32-10 (Salmon et al. 2011): counter-based, so an env's draws for episode k depend only on
// (seed, env, k) -- reproducible regardless of which step the reset happens in or how envs are sharded.
struct Philox {
  uint32_t key[2], ctr[4], out[4];
  int left;
};
__device__ inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ inline void philox_refill(Philox& g) {
  uint32_t c[4] = {g.ctr[0], g.ctr[1], g.ctr[2], g.ctr[3]}, k[2] = {g.key[0], g.key[1]};
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  for (int i = 0; i < 4; ++i) g.out[i] = c[i];
  g.left = 4;
  if (++g.ctr[0] == 0) ++g.ctr[1];
}
__device__ inline Philox philox_init(uint64_t seed, uint32_t env, uint32_t episode) {
  Philox g;
  g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32) ^ (env * 0x9E3779B1u + 0x7F4A7C15u);
  g.ctr[0] = 0; g.ctr[1] = 0; g.ctr[2] = episode; g.ctr[3] = env;
  g.left = 0;
  return g;
}
__device__ inline uint32_t rng_u32(Philox& g) {
  if (g.left == 0) philox_refill(g);
  return g.out[--g.left];
}
__device__ inline double rng_double(Philox& g) {     // [0,1) with 53 random bits (numpy's construction)
  const uint32_t a = rng_u32(g) >> 5, b = rng_u32(g) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
__device__ inline double rng_uniform(Philox& g, double lo, double hi) { return lo + (hi - lo) * rng_double(g); }
__device__ inline int rng_below(Philox& g, int n) { return (int)(((uint64_t)rng_u32(g) * (uint64_t)n) >> 32); }
__device__ inline double rng_normal(Philox& g, double loc, double scale) {   // Box-Muller
  const double u1 = 1.0 - rng_double(g), u2 = rng_double(g);
  return loc + scale * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
// Stream of one dynamic object's domain-randomisation draws: same key as the env's reset stream, counter space
// disjoint from it (word 1 carries a tag + the dynamic slot, word 0 the event: creation or the step of a finish_walk).
__device__ inline Philox philox_object(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot, uint32_t event) {
  Philox g = philox_init(seed, env, episode);
  g.ctr[0] = event; g.ctr[1] = 0x4F424A00u + slot;
  return g;
}
#define DT_OBJ_EVENT_CREATE 0xFFFFFFFFu

// objects.py:384-431 DuckieObj.step.  The reference draws a DuckieObj's domain-randomised parameters from the
// unseeded global np.random (objects.py:349-350, 424-427), so there is no stream to reproduce: on the host path they
// are supplied per env (DTSIM_FIELD_OBJ_PARAMS) and finish_walk takes the non-DR branch; with the device reset
// sampler installed and domain_rand on (`rs`), finish_walk redraws them from the env's Philox stream with the
// reference's distributions: vel = -sign(vel) |N(0.02, 0.005)|, wait = randint(3, 20).
__device__ inline void duckie_step(const SimArrays& A, const DynInit& di, int d, int e, double dt,
                                   const dtsim_reset_sampler* rs, int step_count) {
  const size_t N = A.N, ix = (size_t)d * N + e;
  double time = A.ob_time[ix] + dt;
  A.ob_time[ix] = time;
  if (!A.ob_active[ix]) {
    double w = A.ob_wait[ix] - dt;
    A.ob_wait[ix] = w;
    if (w <= 0) A.ob_active[ix] = 1;
    return;
  }
  double vel = A.ob_vel[ix];
  const double vax = di.heading_x * vel, vaz = di.heading_z * vel;
  const double cx = A.ob_cx[ix] + vax, cz = A.ob_cz[ix] + vaz;
  A.ob_cx[ix] = cx; A.ob_cz[ix] = cz;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A.ob_corners[(size_t)((2 * k) * DTSIM_MAX_DYNAMIC + d) * N + e] += vax;
    A.ob_corners[(size_t)((2 * k + 1) * DTSIM_MAX_DYNAMIC + d) * N + e] += vaz;
  }
  const double ddx = cx - A.ob_sx[ix], ddz = cz - A.ob_sz[ix];
  const double distance = sqrt((ddx * ddx + 0.0) + ddz * ddz);
  double ang = A.ob_angle[ix];
  if (distance > di.walk_distance) {  // finish_walk
    A.ob_sx[ix] = cx; A.ob_sz[ix] = cz;
    ang += 3.141592653589793;
    A.ob_angle[ix] = ang;
    A.ob_active[ix] = 0;
    if (rs != nullptr && rs->domain_rand) {
      Philox g = philox_object(rs->seed, (uint32_t)e, (uint32_t)A.episode[e], (uint32_t)d, (uint32_t)step_count);
      const double mag = fabs(rng_normal(g, 0.02, 0.005));
      A.ob_vel[ix] = vel > 0 ? -mag : (vel < 0 ? mag : -0.0 * mag);   // -1 * np.sign(vel) * |N|
      A.ob_wait[ix] = (double)(3 + rng_below(g, 17));
    } else {
      A.ob_vel[ix] = vel * -1;
      A.ob_wait[ix] = 8;
    }
  }
  const double angle_delta = A.ob_wiggle[ix] * sin(48 * time);
  A.ob_yrot[ix] = (ang + angle_delta) * (180 / 3.141592653589793);
}

// objects.py:531-587 CheckerboardObj.step: a scripted back/forth, left/right, up/down calibration motion of the
// centre, 20/3000 m per call, driven by a step counter that advances by 2 (kept in ob_vel); the collision box is
// never moved (objects.py:507-511 uses the corners of the initial pose).
__device__ inline void checker_step(const SimArrays& A, int d, int e, double dt) {
  const size_t N = A.N, ix = (size_t)d * N + e;
  A.ob_time[ix] += dt;
  const int step = (int)A.ob_vel[ix];
  const double off = 20 * 1.0 / 3000;
  double cx = A.ob_cx[ix], cy = A.ob_cy[ix], cz = A.ob_cz[ix];
  bool move = true;
  if (step < 0) {}
  else if (step < 40) cx += off;
  else if (step < 135) cx -= off;
  else if (step < 170) cx += off;
  else if (step < 200) cz += off;
  else if (step < 260) cz -= off;
  else if (step < 290) cz += off;
  else if (step < 310) cy += off;
  else if (step < 330) cy -= off;
  else if (step < 355) cx -= off;
  else if (step < 370) cy -= off;
  else if (step < 385) cy += off;
  else if (step < 420) cx += off;
  else { cx = A.ob_sx[ix]; cy = 0.0; cz = A.ob_sz[ix]; A.ob_vel[ix] = -20.0; move = false; }
  if (move) A.ob_vel[ix] = (double)(step + 2);
  A.ob_cx[ix] = cx; A.ob_cy[ix] = cy; A.ob_cz[ix] = cz;
}

// objects.py:230-336 DuckiebotObj.step_duckiebot + _update_pos: pure pursuit on the lane curve with
// its own differential-drive kinematics.  Corners are refreshed after a turn, the SAT axes
// (DynInit.norm) deliberately never are (objects.py:333-336 vs :270).
__device__ inline void duckiebot_step(const SimArrays& A, const MapView& m, const DynInit& di, int d, int e, double dt) {
  const size_t N = A.N, ix = (size_t)d * N + e;
  const double px = A.ob_cx[ix], pz = A.ob_cz[ix], ang = A.ob_angle[ix];
  const Lane c0 = lane_pos(m, px, pz, ang);
  if (!c0.in_lane) return;          // the reference raises here; the follower is left where it is
  // per-env parameters (objects.py:198-215: constants, or drawn per instance under domain randomisation)
  const double velocity = A.ob_vel[ix], gain = A.ob_wait[ix], trim = A.ob_wiggle[ix];
  const double follow_dist = A.ob_ext[((size_t)0 * DTSIM_MAX_DYNAMIC + d) * N + e], radius = A.ob_ext[((size_t)1 * DTSIM_MAX_DYNAMIC + d) * N + e];
  const double wheel_dist = A.ob_ext[((size_t)2 * DTSIM_MAX_DYNAMIC + d) * N + e];
  const double kk = 27.0, limit = 1.0;
  double lookup = follow_dist;
  Lane c1;
  c1.in_lane = false;
  for (int it = 0; it < 1000; ++it) {
    c1 = lane_pos(m, c0.pt_x + c0.tan_x * lookup, c0.pt_z + c0.tan_z * lookup, ang);
    if (c1.in_lane) break;
    lookup *= 0.5;
  }
  if (!c1.in_lane) return;
  double vx = c1.pt_x - px, vz = c1.pt_z - pz;
  const double vn = sqrt((vx * vx + 0.0) + vz * vz);
  vx /= vn; vz /= vn;
  const double sa = sin(ang), ca = cos(ang);
  const double dot = (sa * vx + 0.0) + ca * vz;          // right_vec . point_vec
  const double steering = gain * -dot;
  // _update_pos
  const double k_r_inv = (gain + trim) / kk, k_l_inv = (gain - trim) / kk;
  const double omega_r = (velocity + 0.5 * steering * wheel_dist) / radius;
  const double omega_l = (velocity - 0.5 * steering * wheel_dist) / radius;
  const double u_r = omega_r * k_r_inv, u_l = omega_l * k_l_inv;
  const double ur = fmax(fmin(u_r, limit), -limit), ul = fmax(fmin(u_l, limit), -limit);
  if (ul == ur) {
    A.ob_cx[ix] = px + dt * ul * ca;
    A.ob_cz[ix] = pz + dt * ul * -sa;
    return;
  }
  const double w = (ur - ul) / wheel_dist;
  const double r = (wheel_dist * (ul + ur)) / (2 * (ul - ur));
  const double rot = w * dt;
  const double ccx = px + r * sa, ccz = pz + r * ca;
  // graphics.py:254-265 rotate_point
  const double ddx = px - ccx, ddz = pz - ccz;
  const double cr = cos(rot), sr = sin(rot);
  const double npx = ccx + (ddx * cr + ddz * sr), npz = ccz + (ddz * cr - ddx * sr);
  const double nang = ang + rot;
  A.ob_cx[ix] = npx; A.ob_cz[ix] = npz; A.ob_angle[ix] = nang;
  A.ob_yrot[ix] = A.ob_yrot[ix] + rot * 180 / 3.141592653589793;
  // agent_boundbox(pos, robot_width, robot_length, dir, right) collision.py:9-34
  const double ndx = cos(nang), ndz = -sin(nang), nrx = sin(nang), nrz = cos(nang);
  const double hw = 0.5 * A.ob_ext[((size_t)3 * DTSIM_MAX_DYNAMIC + d) * N + e], hl = 0.5 * A.ob_ext[((size_t)4 * DTSIM_MAX_DYNAMIC + d) * N + e];
  const double cx4[4] = {(npx - hw * nrx) - hl * ndx, (npx + hw * nrx) - hl * ndx, (npx + hw * nrx) + hl * ndx, (npx - hw * nrx) + hl * ndx};
  const double cz4[4] = {(npz - hw * nrz) - hl * ndz, (npz + hw * nrz) - hl * ndz, (npz + hw * nrz) + hl * ndz, (npz - hw * nrz) + hl * ndz};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A.ob_corners[(size_t)((2 * k) * DTSIM_MAX_DYNAMIC + d) * N + e] = cx4[k];
    A.ob_corners[(size_t)((2 * k + 1) * DTSIM_MAX_DYNAMIC + d) * N + e] = cz4[k];
  }
}

// duckietown_world DB18 model + SE(2) exponential, restated (PARITY UNPINNED;
// call sites simulator.py:745-755, 2076-2088).  Mirrors oracle/sim.py DynamicsDB18.
struct Dyn { double x, y, c, s, u, w; };

__device__ inline void dyn_integrate(Dyn& q, double dt, double L_, double R_, double war, double wal) {
  const double R = fmin(fmax(R_, -1.0), 1.0), L = fmin(fmax(L_, -1.0), 1.0);
  const double u1 = 5.0, u2 = 0.0, u3 = 0.0, w1 = 4.0, w2 = 0.0, w3 = 0.0, uar = 1.5, ual = 1.5;
  double u = q.u, w = q.w;
  const double acc_u = ((-u1 * u - u2 * w) + u3 * w * w) + (uar * R + ual * L);
  const double acc_w = ((-w1 * w - w2 * u) - w3 * u * w) + (war * R + (-wal) * L);
  u = u + dt * acc_u;
  w = w + dt * acc_w;
  const double th = dt * w, vx = dt * u;
  double cd, sd, tx, ty;
  if (fabs(th) < 1e-12) { cd = 1.0; sd = th; tx = vx; ty = 0.0; }
  else {
    sd = sin(th); cd = cos(th);
    const double Aa = sd / th, Bb = (1.0 - cd) / th;
    tx = Aa * vx; ty = Bb * vx;
  }
  const double c0 = q.c, s0 = q.s;
  q.x = (c0 * tx + (-s0) * ty) + q.x;
  q.y = (s0 * tx + c0 * ty) + q.y;
  q.c = c0 * cd + (-s0) * sd;
  q.s = s0 * cd + c0 * sd;
  q.u = u; q.w = w;
}

// DTSIM_F_LIGHT_CAPTURE: the model-view the env's last frame left behind -- _render_img's Rx(cam_angle) T(0, 0, forward) LookAt (simulator.py:1758-1803) at the
// pose the episode ended at -- as eye C, yaw (sa, ca), pitch (sth, cth); read BEFORE the reset overwrites the pose.  gl_light_to_eye of the gym facade, on the device.
struct LightMV { double Cx, Cy, Cz, sa, ca, sth, cth; };
__device__ inline LightMV light_mv(const SimArrays& A, int e, int domain_rand) {
  const size_t N = A.N;
  const double a = A.angle[e], sa = sin(a), ca = cos(a), th = (double)A.cam[1 * N + e];
  double px = A.pos_x[e], py = 0.0, pz = A.pos_z[e];
  if (domain_rand) { px += (double)A.cam[3 * N + e]; py += (double)A.cam[4 * N + e]; pz += (double)A.cam[5 * N + e]; }   // simulator.py:1768-1769
  return LightMV{px + DT_CAMERA_FORWARD_DIST * ca, py + (double)A.cam[0 * N + e], pz - DT_CAMERA_FORWARD_DIST * sa, sa, ca, sin(th), cos(th)};
}
__device__ inline void light_through(const SimArrays& A, int e, const LightMV& mv) {   // the light apply_init just wrote -> eye space of `mv`
  const size_t N = A.N;
  const double w = (double)A.colors[15 * N + e];
  double rx = (double)A.colors[12 * N + e], ry = (double)A.colors[13 * N + e], rz = (double)A.colors[14 * N + e];
  if (w != 0.0) { rx = rx / w - mv.Cx; ry = ry / w - mv.Cy; rz = rz / w - mv.Cz; }
  const double xla = rx * mv.sa + rz * mv.ca, zla = -(rx * mv.ca - rz * mv.sa);
  A.colors[12 * N + e] = (float)xla;
  A.colors[13 * N + e] = (float)(ry * mv.cth - zla * mv.sth);
  A.colors[14 * N + e] = (float)(ry * mv.sth + zla * mv.cth);
  A.colors[15 * N + e] = w != 0.0 ? 1.f : 0.f;
}

// Simulator.reset()'s hand-over of one env (simulator.py:740-755) from a host-drawn state.
__device__ inline void apply_init(const SimArrays& A, const MapSet& M, int e, const dtsim_init_state& st,
                                  int delay_steps, const dtsim_reset_sampler* rs = nullptr) {
  const size_t N = A.N;
  const int new_map = st.map_id & ~DTSIM_MAP_RELOAD;
  const bool reload = (st.map_id & DTSIM_MAP_RELOAD) != 0;
  const MapHdr* mh = reinterpret_cast<const MapHdr*>(M.blobs + M.blob_off[new_map]);
  A.pos_x[e] = st.pos[0]; A.pos_z[e] = st.pos[2]; A.angle[e] = st.angle;
  // cartesian_from_weird simulator.py:1629-1638
  A.q_x[e] = st.pos[0];
  A.q_y[e] = mh->grid_h * mh->tile_size - st.pos[2];
  A.q_c[e] = cos(st.angle); A.q_s[e] = sin(st.angle);
  A.vel_u[e] = 0.0; A.vel_w[e] = 0.0;
  for (int k = 0; k < DTSIM_MAX_DELAY * 2; ++k) A.ring[(size_t)k * N + e] = 0.0;
  A.ring_head[e] = 0;
  if (st.dynamics_trim_on) { A.war[e] = 15.0 * (1.0 + st.dynamics_trim); A.wal[e] = 15.0 * (1.0 - st.dynamics_trim); }
  else { A.war[e] = 15.0; A.wal[e] = 15.0; }
  A.wheel_dist[e] = st.wheel_dist;
  A.step_count[e] = 0; A.timestamp[e] = 0.0; A.speed[e] = 0.0;
  A.reward[e] = 0.0; A.done[e] = 0; A.done_code[e] = DTSIM_DONE_IN_PROGRESS;
  A.wheels[e] = 0.0; A.wheels[N + e] = 0.0;
  A.cam[0 * N + e] = (float)st.cam_height;
  A.cam[1 * N + e] = (float)(st.cam_angle_deg * (3.141592653589793 / 180.0));
  A.cam[2 * N + e] = (float)(st.cam_fov_y_deg * (3.141592653589793 / 180.0));
  A.cam[3 * N + e] = (float)st.camera_noise[0];
  A.cam[4 * N + e] = (float)st.camera_noise[1];
  A.cam[5 * N + e] = (float)st.camera_noise[2];
  for (int k = 0; k < 3; ++k) {
    A.colors[(0 + k) * N + e] = (float)st.horizon_color[k];
    A.colors[(3 + k) * N + e] = (float)st.ground_color[k];
    A.colors[(6 + k) * N + e] = (float)st.light_ambient[k];
    A.colors[(9 + k) * N + e] = (float)st.light_diffuse[k];
  }
  for (int k = 0; k < 4; ++k) A.colors[(12 + k) * N + e] = (float)st.light_pos[k];
  // World objects are created at map load and persist across resets in the reference
  // (simulator.py:349,865); (re)create them only when the env's map changes.
  if (A.map_id[e] != new_map || reload) {
    const DynInit* dyn = M.dyn + (size_t)new_map * DTSIM_MAX_DYNAMIC;
    for (int d = 0; d < mh->n_dyn; ++d) {
      const size_t ix = (size_t)d * N + e;
      A.ob_cx[ix] = dyn[d].cx; A.ob_cz[ix] = dyn[d].cz;
      A.ob_sx[ix] = dyn[d].cx; A.ob_sz[ix] = dyn[d].cz; A.ob_cy[ix] = 0.0;
      for (int k = 0; k < 8; ++k) A.ob_corners[(size_t)(k * DTSIM_MAX_DYNAMIC + d) * N + e] = dyn[d].corners[k];
      A.ob_vel[ix] = dyn[d].vel; A.ob_wait[ix] = dyn[d].wait_time; A.ob_time[ix] = 0.0;
      A.ob_angle[ix] = dyn[d].angle; A.ob_wiggle[ix] = dyn[d].wiggle;
      if (rs != nullptr && dyn[d].kind == 1) {       // DuckieObj.__init__ draws (objects.py:348-363), device sampler only
        Philox g = philox_object(rs->seed, (uint32_t)e, (uint32_t)A.episode[e], (uint32_t)d, DT_OBJ_EVENT_CREATE);
        if (rs->domain_rand) {
          A.ob_wait[ix] = (double)(3 + rng_below(g, 17));                 // np.random.randint(3, 20)
          A.ob_vel[ix] = fabs(rng_normal(g, 0.02, 0.005));                // np.abs(np.random.normal(0.02, 0.005))
        }
        A.ob_wiggle[ix] = 3.141592653589793 / (double)(14 + rng_below(g, 3));   // np.pi / choice([14, 15, 16]): always drawn
      }
      {  // DuckiebotObj.__init__ (objects.py:198-215): DynInit carries follow_dist / velocity / gain / trim of the non-DR branch
        double ext[5] = {dyn[d].walk_distance, 0.0318, 0.102, DT_ROBOT_WIDTH, DT_ROBOT_LENGTH};
        if (rs != nullptr && rs->domain_rand && dyn[d].kind == 2) {   // the DR branch, device sampler only (np.random there: unseeded)
          Philox g = philox_object(rs->seed, (uint32_t)e, (uint32_t)A.episode[e], (uint32_t)d, DT_OBJ_EVENT_CREATE);
          ext[0] = rng_uniform(g, 0.3, 0.4);                              // follow_dist
          A.ob_vel[ix] = rng_uniform(g, 0.05, 0.15);                      // velocity
          A.ob_wait[ix] = dyn[d].wait_time + rng_uniform(g, -0.3, 0.3);   // gain + U(-0.3, 0.3)
          A.ob_wiggle[ix] = dyn[d].wiggle + rng_uniform(g, -0.1, 0.1) + 2;   // trim + U(-0.1, 0.1) + 2   (objects.py:203, sic)
          ext[1] = 0.0318 + 0.0002 * rng_uniform(g, -1.0, 1.0);
          ext[2] = 0.102 + 0.01 * rng_uniform(g, -1.0, 1.0);
          ext[3] = DT_ROBOT_WIDTH + 0.01 * rng_uniform(g, -1.0, 1.0);
          ext[4] = DT_ROBOT_LENGTH + 0.01 * rng_uniform(g, -1.0, 1.0);
        }
        for (int k = 0; k < 5; ++k) A.ob_ext[((size_t)k * DTSIM_MAX_DYNAMIC + d) * N + e] = ext[k];
      }
      A.ob_yrot[ix] = dyn[d].angle * (180 / 3.141592653589793);
      A.ob_active[ix] = 0;
    }
    for (int o = 0; o < DTSIM_MAX_OBJECTS; ++o) A.ob_visible[(size_t)o * N + e] = 1;
    {  // TrafficLightObj.__init__ (objects.py:441-453): object clock 0, initial pattern
      const double* objs = reinterpret_cast<const double*>(M.blobs + M.blob_off[new_map]) + mh->off_objs;
      for (int o = 0; o < DTSIM_MAX_OBJECTS; ++o)
        A.ob_light[(size_t)o * N + e] = (o < mh->n_obj && objs[o * OBJ_WORDS + 4] > 0.0) ? (uint8_t)objs[o * OBJ_WORDS + 5] : (uint8_t)0;
      A.tl_time[e] = 0.0;
    }
    A.map_id[e] = new_map;
  }
  // info fields of the new pose are filled by the first step / by k_reset's tail
  (void)delay_steps;
}

// _inconvenient_spawn simulator.py:1461-1471 (uses obj.pos; for a DuckieObj that is its current centre,
// objects.py:408)
__device__ inline bool inconvenient_spawn(const MapView& m, const SimArrays& A, int e, double px, double pz) {
  bool inc = false;
  const int N = A.N;
  for (int o = 0; o < m.h->n_obj; ++o) {
    if (!A.ob_visible[(size_t)o * N + e]) continue;
    const double* ob = m.objs + o * OBJ_WORDS;
    double ox = ob[0], oy = 0.0, oz = ob[1];
    const int slot = (int)ob[3];
    if (slot >= 0) { ox = A.ob_cx[(size_t)slot * N + e]; oy = A.ob_cy[(size_t)slot * N + e]; oz = A.ob_cz[(size_t)slot * N + e]; }
    const double ddx = ox - px, ddz = oz - pz;
    inc = inc || (sqrt((ddx * ddx + oy * oy) + ddz * ddz) < ob[2]);
  }
  return inc;
}

// ---- device-side reset sampler (dtsim_reset_sampler): Philox helpers are defined above (objects use them too)
// _perturb simulator.py:1065-1085: val * U(1 - scale, 1 + scale) per component
__device__ inline void perturb3(Philox& g, bool on, const double v[3], double scale, double out[3]) {
  for (int k = 0; k < 3; ++k) out[k] = on ? v[k] * rng_uniform(g, 1.0 - scale, 1.0 + scale) : v[k];
}

// What Simulator.reset() decides for one env (simulator.py:546-738), drawn on the device: the same
// distributions and acceptance test as dtsim/reset.py + the host spawn loop, from the Philox stream.
__device__ inline dtsim_init_state sample_init(const SimArrays& A, const MapSet& M, const uint64_t* blobs,
                                               const dtsim_reset_sampler& rs, int e, int episode, int map_id) {
  const int N = A.N;
  Philox g = philox_init(rs.seed, (uint32_t)e, (uint32_t)episode);
  dtsim_init_state st;
  st.map_id = map_id;
  const bool dr = rs.domain_rand != 0;
  // randomizer.py:36-91 (all seven keys are always drawn) -- sorted-key order kept for readability only
  const double cam_angle_f = rng_uniform(g, 0.8, 1.2), cam_fov_f = rng_uniform(g, 0.8, 1.2), cam_height_f = rng_uniform(g, 0.92, 1.08);
  for (int k = 0; k < 3; ++k) st.camera_noise[k] = rng_uniform(g, -0.005, 0.005);
  const int horz_mode = rng_below(g, 4);
  const double lp[3] = {rng_uniform(g, -150, 150), rng_uniform(g, 170, 220), rng_uniform(g, -150, 150)};
  const double trim = rng_normal(g, 0.0, 0.02);
  const double sky[3] = {rs.color_sky[0], rs.color_sky[1], rs.color_sky[2]};
  const double wall[3] = {0.64, 0.71, 0.28}, dark[3] = {0.15, 0.15, 0.15}, light[3] = {0.9, 0.9, 0.9};
  if (dr) {                                          // simulator.py:551-571
    if (horz_mode == 0) perturb3(g, true, sky, 0.1, st.horizon_color);
    else if (horz_mode == 1) perturb3(g, true, wall, 0.1, st.horizon_color);
    else if (horz_mode == 2) perturb3(g, true, dark, 0.4, st.horizon_color);
    else perturb3(g, true, light, 0.4, st.horizon_color);
    st.light_pos[0] = lp[0]; st.light_pos[1] = lp[1]; st.light_pos[2] = lp[2]; st.light_pos[3] = 0.0;
  } else {
    for (int k = 0; k < 3; ++k) st.horizon_color[k] = sky[k];
    st.light_pos[0] = 0.0; st.light_pos[1] = 3.0; st.light_pos[2] = 0.0; st.light_pos[3] = 1.0;
  }
  const double amb[3] = {0.25, 0.25, 0.25}, dif[3] = {0.35, 0.35, 0.35};
  const double gnd[3] = {rs.color_ground[0], rs.color_ground[1], rs.color_ground[2]};
  perturb3(g, dr, amb, 0.3, st.light_ambient);
  perturb3(g, dr, dif, 0.99, st.light_diffuse);
  perturb3(g, dr, gnd, 0.3, st.ground_color);
  st.wheel_dist = dr ? 0.102 * rng_uniform(g, 0.9, 1.1) : 0.102;
  st.cam_height = 0.108 * (dr ? cam_height_f : 1.0);
  st.cam_angle_deg = 19.15 * (dr ? cam_angle_f : 1.0);
  st.cam_fov_y_deg = 75.0 * (dr ? cam_fov_f : 1.0);
  st.dynamics_trim_on = rs.dynamics_rand ? 1 : 0;
  st.dynamics_trim = trim;
  const MapView m = map_view(blobs + M.blob_off[map_id]);
  const DynInit* dyn = M.dyn + (size_t)map_id * DTSIM_MAX_DYNAMIC;
  // optional objects: visible with probability 1/2 under domain randomisation (simulator.py:653-656)
  for (int o = 0; o < m.h->n_obj; ++o) {
    const bool optional = m.objs[o * OBJ_WORDS + 3] < -1.5;       // dyn_slot word: -2 marks an optional static object
    A.ob_visible[(size_t)o * N + e] = (optional && dr) ? (uint8_t)(rng_below(g, 2) == 0) : (uint8_t)1;
  }
  // start tile (simulator.py:659-676)
  int ti = rs.start_tile[map_id][0], tj = rs.start_tile[map_id][1];
  if (ti < 0) {
    const int n_tiles = m.h->grid_w * m.h->grid_h;
    for (int tries = 0; tries < 4096; ++tries) {     // uniform over the drivable tiles by rejection
      const int t = rng_below(g, n_tiles);
      if (m.tiles[t].drivable) { ti = t % m.h->grid_w; tj = t / m.h->grid_w; break; }
    }
    if (ti < 0) { ti = 0; tj = 0; }
  }
  // spawn loop (simulator.py:692-738)
  const double ts = m.h->tile_size, M_deg = rs.accept_start_angle_deg;
  st.pos[0] = 1.0; st.pos[1] = 0.0; st.pos[2] = 1.0; st.angle = 1.0;                 // fallback pose :732-736
  if (rs.has_start_pose[map_id]) {                   // the map specifies a starting pose (simulator.py:679-688): no rejection loop
    st.pos[0] = ti * ts + rs.start_pose[map_id][0]; st.pos[2] = tj * ts + rs.start_pose[map_id][1];
    st.angle = rs.start_pose[map_id][2];
    return st;
  }
  for (int a = 0; a < rs.max_attempts; ++a) {
    const double x = rng_uniform(g, ti, ti + 1) * ts, z = rng_uniform(g, tj, tj + 1) * ts;
    const double ang = rng_uniform(g, 0.0, 6.283185307179586);
    if (inconvenient_spawn(m, A, e, x, z)) continue;
    if (!valid_pose(m, A, dyn, e, x, z, ang, 1.3, nullptr)) continue;
    const Lane L = lane_pos(m, x, z, ang);
    if (!L.in_lane) continue;
    if (!(-M_deg < L.angle_deg && L.angle_deg < M_deg)) continue;
    st.pos[0] = x; st.pos[2] = z; st.angle = ang;
    break;
  }
  return st;
}

// Fill tile / lane / prox info for the current pose (what get_agent_info reports,
// simulator.py:1586-1627).
__device__ inline void fill_info(const SimArrays& A, const MapView& m, const DynInit* dyn, int e) {
  const size_t N = A.N;
  const double px = A.pos_x[e], pz = A.pos_z[e], ang = A.angle[e];
  int i, j;
  tile_at(m, px, pz, i, j);
  A.tile_i[e] = i; A.tile_j[e] = j;
  const Lane L = lane_pos(m, px, pz, ang);
  A.in_lane[e] = L.in_lane;
  A.lane[0 * N + e] = L.dist; A.lane[1 * N + e] = L.dot_dir;
  A.lane[2 * N + e] = L.angle_deg; A.lane[3 * N + e] = L.angle_rad;
  A.prox[e] = proximity(m, A, dyn, e, px, pz, ang);
}

// Stage every map blob into LDS (cooperative, 8-byte words).
__device__ inline const uint64_t* stage_maps(const MapSet& M, uint64_t* lds) {
  for (int k = threadIdx.x; k < M.total_words; k += blockDim.x) lds[k] = M.blobs[k];
  __syncthreads();
  return lds;
}

// SAMPLER: the device-side reset sampler is compiled in (its rejection loop and Philox state cost registers, so
// the pool / no-auto-reset launches use the lean instantiation).
// Which map a device-sampled reset moves the env to: MultiMapEnv's round robin (multimap_env.py:44-49), or
// randomize_maps_on_reset's uniform draw -- `self.np_random.choice(self.map_names)`, the first draw of reset()
// (simulator.py:541-544) -- from its own counter word of the env's Philox key.
__device__ inline int sampler_next_map(const dtsim_reset_sampler& rs, int n_maps, int cur, int e, int episode) {
  if (rs.map_cycle == 1) return (cur + 1) % n_maps;
  if (rs.map_cycle == 2) {
    Philox g = philox_init(rs.seed, (uint32_t)e, (uint32_t)episode);
    g.ctr[1] = 0x4D415000u;
    return rng_below(g, n_maps);
  }
  return cur;
}

template <bool SAMPLER, int L>
__global__ __launch_bounds__(STEP_BLOCK) void k_step(SimArrays A, MapSet M, StepParams P, const void* actions,
                                                     const dtsim_init_state* pool) {
  extern __shared__ uint64_t lds[];
  const uint64_t* blobs = stage_maps(M, lds);
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = gt / L;                              // L adjacent lanes per env (Coop above); L = 1: one thread per env
  const Coop co{gt % L, L};
  const int N = A.N;
  if (e >= N) return;
  const double dt = P.delta_time;
  const bool pose_only = (P.step_flags & DTSIM_STEP_POSE_ONLY) != 0;     // `_update_pos` (simulator.py:2076-2088)
  // update_physics (simulator.py:1551) and _update_pos (:2076) take the wheel pair as given: the (vel, steering)
  // kinematics are DuckietownEnv.step's (envs/duckietown_env.py:36-61) and np.clip is Simulator.step's (:1670)
  const bool raw_wheels = (P.step_flags & (DTSIM_STEP_ONE_UPDATE | DTSIM_STEP_POSE_ONLY)) != 0;
  const int n_updates = (P.step_flags & (DTSIM_STEP_ONE_UPDATE | DTSIM_STEP_POSE_ONLY)) ? 1 : P.frame_skip;

  for (int s = 0; s < P.n_steps; ++s) {
    // ---- auto reset (DTSIM_F_AUTO_RESET): the caller's reset() after a done=True
    if (P.auto_reset && !pose_only && A.done[e]) {
      const int ep = A.episode[e] + 1;
      const LightMV mv = light_mv(A, e, P.domain_rand);          // (the pose the episode ended at: what the last frame was drawn from)
      A.episode[e] = ep;
      if (SAMPLER) {                                 // device-side sampling (takes precedence over the pool)
        const int cur = A.map_id[e];
        const int nm = sampler_next_map(*P.sampler, M.n_maps, cur, e, ep);
        const bool reload = P.sampler->map_cycle == 2;
        // the objects of the new (or reloaded) map must exist before the spawn test looks at them
        if (cur != nm || reload) { dtsim_init_state z{}; z.map_id = nm | (reload ? DTSIM_MAP_RELOAD : 0); z.wheel_dist = 0.102; apply_init(A, M, e, z, P.delay_steps, P.sampler); }
        const dtsim_init_state st = sample_init(A, M, blobs, *P.sampler, e, ep, nm);
        apply_init(A, M, e, st, P.delay_steps, P.sampler);
      } else {
        const long long slot = ((long long)e + (long long)ep * N) % P.n_pool;
        apply_init(A, M, e, pool[slot], P.delay_steps);
      }
      if (P.light_capture) light_through(A, e, mv);
    }
    const int mid = A.map_id[e];
    const MapView m = map_view(blobs + M.blob_off[mid]);
    const DynInit* dyn = M.dyn + (size_t)mid * DTSIM_MAX_DYNAMIC;

    // ---- action (envs/duckietown_env.py:36-61, simulator.py:1670)
    double a0, a1;
    const size_t aoff = ((size_t)s * N + e) * 2;
    if (P.actions_f64) { a0 = ((const double*)actions)[aoff]; a1 = ((const double*)actions)[aoff + 1]; }
    else { a0 = (double)((const float*)actions)[aoff]; a1 = (double)((const float*)actions)[aoff + 1]; }
    double left, right;
    if (P.action_mode == DTSIM_ACTION_VEL_STEER && !raw_wheels) {
      const double vel = a0, steer = a1, baseline = A.wheel_dist[e];
      const double k_r_inv = (P.gain + P.trim) / P.k, k_l_inv = (P.gain - P.trim) / P.k;
      const double omega_r = (vel + 0.5 * steer * baseline) / P.radius;
      const double omega_l = (vel - 0.5 * steer * baseline) / P.radius;
      const double u_r = omega_r * k_r_inv, u_l = omega_l * k_l_inv;
      right = fmax(fmin(u_r, P.limit), -P.limit);
      left = fmax(fmin(u_l, P.limit), -P.limit);
    } else { left = a0; right = a1; }
    if (!raw_wheels) {
      left = fmin(fmax(left, -1.0), 1.0);   // np.clip(action, -1, 1) simulator.py:1670 (Simulator.step only)
      right = fmin(fmax(right, -1.0), 1.0);
    }
    if (!pose_only) { A.wheels[e] = left; A.wheels[(size_t)N + e] = right; }

    // ---- frame_skip x update_physics (simulator.py:1551-1584)
    Dyn q = {A.q_x[e], A.q_y[e], A.q_c[e], A.q_s[e], A.vel_u[e], A.vel_w[e]};
    double px = A.pos_x[e], pz = A.pos_z[e], ang = A.angle[e];
    const double war = A.war[e], wal = A.wal[e];
    int head = A.ring_head[e];
    int sc = A.step_count[e];
    double ts_ = A.timestamp[e], speed = A.speed[e];
    const double Hts = m.h->grid_h * m.h->tile_size;
    for (int f = 0; f < n_updates; ++f) {
      double l_use = left, r_use = right;
      if (P.delay_steps > 0) {  // ApplyDelay: command issued delay_steps ago
        l_use = A.ring[(size_t)(2 * head) * N + e];
        r_use = A.ring[(size_t)(2 * head + 1) * N + e];
        A.ring[(size_t)(2 * head) * N + e] = left;
        A.ring[(size_t)(2 * head + 1) * N + e] = right;
        head = (head + 1) % P.delay_steps;
      }
      dyn_integrate(q, dt, l_use, r_use, war, wal);
      // weird_from_cartesian simulator.py:1640-1652
      const double nx = q.x, nz = Hts - q.y;
      const double ddx = nx - px, ddz = nz - pz;
      px = nx; pz = nz; ang = atan2(q.s, q.c);
      if (pose_only) continue;                         // `_update_pos` stops here
      sc += 1; ts_ += dt;
      speed = sqrt((ddx * ddx + 0.0) + ddz * ddz) / dt;
      for (int d = co.sub; d < m.h->n_dyn; d += L) {  // simulator.py:1571-1584; one object per lane of the env's group
        if (dyn[d].kind == 2) duckiebot_step(A, m, dyn[d], d, e, dt);
        else if (dyn[d].kind == 3) checker_step(A, d, e, dt);
        else duckie_step(A, dyn[d], d, e, dt, SAMPLER ? P.sampler : nullptr, sc);
      }
      if (m.h->n_lights > 0) {                         // TrafficLightObj.step (objects.py:455-463)
        const double tl = A.tl_time[e] + dt;
        A.tl_time[e] = tl;
        // round(time, 3) % freq == 0  <=>  the time rounded to milliseconds is a whole multiple of freq seconds
        const long long ms = llrint(tl * 1000.0);
        for (int o = 0; o < m.h->n_obj; ++o) {
          const long long f_ms = (long long)m.objs[o * OBJ_WORDS + 4] * 1000;
          if (f_ms > 0 && ms % f_ms == 0) A.ob_light[(size_t)o * N + e] ^= 1;
        }
      }
    }
    A.q_x[e] = q.x; A.q_y[e] = q.y; A.q_c[e] = q.c; A.q_s[e] = q.s; A.vel_u[e] = q.u; A.vel_w[e] = q.w;
    A.pos_x[e] = px; A.pos_z[e] = pz; A.angle[e] = ang;
    A.ring_head[e] = head; A.step_count[e] = sc; A.timestamp[e] = ts_; A.speed[e] = speed;
    if (pose_only) continue;

    // ---- _compute_done_reward (simulator.py:1685-1705)
    int ti, tj;
    tile_at(m, px, pz, ti, tj);
    A.tile_i[e] = ti; A.tile_j[e] = tj;
    const Lane Ln = lane_pos(m, px, pz, ang);
    if (L > 1) __threadfence_block();                // the objects were stepped by different lanes of the group
    const double prox = proximity(m, A, dyn, e, px, pz, ang, co);
    A.in_lane[e] = Ln.in_lane;
    A.lane[e] = Ln.dist; A.lane[(size_t)N + e] = Ln.dot_dir;
    A.lane[(size_t)2 * N + e] = Ln.angle_deg; A.lane[(size_t)3 * N + e] = Ln.angle_rad;
    A.prox[e] = prox;
    double reward; uint8_t done, code;
    if (!valid_pose(m, A, dyn, e, px, pz, ang, 1.0, nullptr, co)) {
      reward = DT_REWARD_INVALID_POSE; done = 1; code = DTSIM_DONE_INVALID_POSE;
    } else if (sc >= P.max_steps) {
      reward = 0.0; done = 1; code = DTSIM_DONE_MAX_STEPS;
    } else {
      reward = compute_reward(Ln, prox, P.robot_speed); done = 0; code = DTSIM_DONE_IN_PROGRESS;
    }
    A.reward[e] = reward; A.done[e] = done; A.done_code[e] = code;
  }
}

__global__ __launch_bounds__(STEP_BLOCK) void k_reset(SimArrays A, MapSet M, StepParams P, const uint8_t* mask,
                                                      const dtsim_init_state* states) {
  extern __shared__ uint64_t lds[];
  const uint64_t* blobs = stage_maps(M, lds);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A.N) return;
  if (mask != nullptr && !mask[e]) return;
  if (states == nullptr) {                           // device sampler (dtsim_reset(h, mask, NULL))
    const int ep = A.episode[e] + 1;
    const bool had_frame = A.map_id[e] >= 0;          // (the env's very first reset: nothing drawn yet, the model-view is the identity)
    const LightMV mv = had_frame ? light_mv(A, e, P.domain_rand) : LightMV{};
    A.episode[e] = ep;
    const int cur = A.map_id[e] < 0 ? (P.sampler->map_cycle ? e % M.n_maps : 0) : A.map_id[e];
    const bool reload = P.sampler->map_cycle == 2;
    const int nm = reload ? sampler_next_map(*P.sampler, M.n_maps, cur, e, ep)
                          : ((A.map_id[e] >= 0 && P.sampler->map_cycle) ? (cur + 1) % M.n_maps : cur);
    // objects of a fresh env must exist before the spawn test looks at them
    if (A.map_id[e] != nm || reload) { dtsim_init_state z{}; z.map_id = nm | (reload ? DTSIM_MAP_RELOAD : 0); z.wheel_dist = 0.102; apply_init(A, M, e, z, P.delay_steps, P.sampler); }
    const dtsim_init_state st = sample_init(A, M, blobs, *P.sampler, e, ep, nm);
    apply_init(A, M, e, st, P.delay_steps, P.sampler);
    if (P.light_capture && had_frame) light_through(A, e, mv);
  } else
  apply_init(A, M, e, states[e], P.delay_steps);
  const int mid = A.map_id[e];
  fill_info(A, map_view(blobs + M.blob_off[mid]), M.dyn + (size_t)mid * DTSIM_MAX_DYNAMIC, e);
}

__global__ __launch_bounds__(STEP_BLOCK) void k_query(SimArrays A, MapSet M, StepParams P, int n,
                                                      const int32_t* env_idx, const double* poses,
                                                      double sf, dtsim_probe* out) {
  extern __shared__ uint64_t lds[];
  const uint64_t* blobs = stage_maps(M, lds);
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= n) return;
  const int e = env_idx[qi];
  const int mid = A.map_id[e];
  const MapView m = map_view(blobs + M.blob_off[mid]);
  const DynInit* dyn = M.dyn + (size_t)mid * DTSIM_MAX_DYNAMIC;
  const double px = poses[3 * qi], pz = poses[3 * qi + 1], ang = poses[3 * qi + 2];
  dtsim_probe r;
  int i, j;
  const TileRec* t = tile_at(m, px, pz, i, j);
  r.tile_i = i; r.tile_j = j;
  r.drivable = (t != nullptr && t->drivable);
  const double ca = cos(ang), sa = sin(ang);
  double ac[8];
  agent_corners(px, pz, ca, -sa, sa, ca, ac);
  r.collision = collision(m, A, dyn, e, ac, ca, -sa, sa, ca);
  r.valid = valid_pose(m, A, dyn, e, px, pz, ang, sf, nullptr);
  const Lane L = lane_pos(m, px, pz, ang);
  r.in_lane = L.in_lane; r.curve_idx = L.curve_idx; r.t = L.t;
  r.point[0] = L.pt_x; r.point[1] = L.pt_z; r.tangent[0] = L.tan_x; r.tangent[1] = L.tan_z;
  r.dist = L.dist; r.dot_dir = L.dot_dir; r.angle_deg = L.angle_deg; r.angle_rad = L.angle_rad;
  r.prox = proximity(m, A, dyn, e, px, pz, ang);
  r.reward = compute_reward(L, r.prox, P.robot_speed);
  const bool inc = inconvenient_spawn(m, A, e, px, pz);
  r.inconvenient = inc;
  r.pad[0] = r.pad[1] = r.pad[2] = 0;
  out[qi] = r;
}

}  // namespace

void dt_launch_step(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P,
                    const void* actions, const dtsim_init_state* pool) {
  // P.lanes adjacent lanes per env (1, 2, 4 or 8; Coop in this file): more wavefronts for the same envs, object loops split
  const int L = (P.lanes == 2 || P.lanes == 4 || P.lanes == 8) ? P.lanes : 1;
  const int grid = (int)(((long long)A.N * L + STEP_BLOCK - 1) / STEP_BLOCK);
  const size_t lds = (size_t)M.total_words * 8;
#define LAUNCH_STEP(S_, L_) hipLaunchKernelGGL((k_step<S_, L_>), dim3(grid), dim3(STEP_BLOCK), lds, s, A, M, P, actions, pool)
  if (P.sampler) { if (L == 8) LAUNCH_STEP(true, 8); else if (L == 4) LAUNCH_STEP(true, 4); else if (L == 2) LAUNCH_STEP(true, 2); else LAUNCH_STEP(true, 1); }
  else { if (L == 8) LAUNCH_STEP(false, 8); else if (L == 4) LAUNCH_STEP(false, 4); else if (L == 2) LAUNCH_STEP(false, 2); else LAUNCH_STEP(false, 1); }
#undef LAUNCH_STEP
}

void dt_launch_reset(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P,
                     const uint8_t* mask, const dtsim_init_state* states) {
  const int grid = (A.N + STEP_BLOCK - 1) / STEP_BLOCK;
  hipLaunchKernelGGL(k_reset, dim3(grid), dim3(STEP_BLOCK), (size_t)M.total_words * 8, s, A, M, P, mask, states);
}

void dt_launch_query(hipStream_t s, const SimArrays& A, const MapSet& M, const StepParams& P, int n,
                     const int32_t* env_idx, const double* poses, double safety_factor, dtsim_probe* out) {
  const int grid = (n + STEP_BLOCK - 1) / STEP_BLOCK;
  hipLaunchKernelGGL(k_query, dim3(grid), dim3(STEP_BLOCK), (size_t)M.total_words * 8, s, A, M, P, n, env_idx,
                     poses, safety_factor, out);
}
