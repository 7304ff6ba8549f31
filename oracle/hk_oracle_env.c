/*
 * hk_oracle_env.c — CPU ORACLE (test infrastructure): one Unity FixedUpdate tick of the racing environment,
 * components a4-a11 of SURVEY §8, restated from the reference C# line by line (paths relative to
 * /root/reference/Assets/Karting/Scripts/):
 *   HKA = AI/HierarchicalKartAgent.cs   KA = AI/KartAgent.cs   AK = KartSystems/ArcadeKart.cs
 *   REC = RacingEnvController.cs        DPT = DiscretePositionTracker.cs
 * The closed-source engine (Unity 2020.3 / PhysX 4.1: pose integration, contacts, raycasts, trigger dispatch) is
 * RE-STATED ANALYTICALLY in 2-D (see DESIGN.md "Engine restatement"): parity with Unity at that boundary is unpinned.
 * Geometry queries here are BRUTE FORCE over every wall / trigger / kart, on purpose: the HIP kernels use culled
 * candidate lists and must still agree bit for bit.
 *
 * Canonical script order inside a tick (all scripts have executionOrder 0 in the reference):
 *   (a) REC.FixedUpdate  (b) every agent: KA.FixedUpdate -> HKA.FixedUpdate (SolveLQR, planFixed)
 *   (c) every kart: ArcadeKart.FixedUpdate (MoveVehicle)  (d) engine: integrate, kart-kart, kart-wall, triggers.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hk_oracle.h"
#include "../include/hk_detmath.h"
#include "hk_oracle_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

enum { XI = 0, ZI = 1, VI = 2, HI = 3 };

#define DEG2RAD_F 0.0174532924f      /* Mathf.Deg2Rad */
#define TWO_PI_F (2.0f * HK_PI_F)    /* 2 * Mathf.PI, evaluated in float (HKA:735) */
#define KART_CAP_R 0.45f             /* BaseKartClassic.prefab capsule radius */
#define KART_CAP_Z0 (-0.107f - 0.55f)/* capsule core segment, kart-local z: centre -0.107, half length h/2 - r = 0.55 */
#define KART_CAP_Z1 (-0.107f + 0.55f)
#define SENSOR_LZ 0.1f               /* MLAgent_Sensors origin, kart-local (0, 0.5, 0.1) */
#define SENSOR_LY 0.5f
#define CAP_CENTER_LY 0.582f         /* capsule centre height above the kart origin */
#define TRIG_HX 5.0f                 /* Trigger box half extents (10 x 1 x 1) */
#define TRIG_HZ 0.5f

/* ------------------------------------------------------------------ small float helpers */
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }
static inline float f_abs(float a) { return a < 0.0f ? -a : a; }
static inline float f_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); } /* Mathf.Clamp */
static inline float f_sign(float v) { return v >= 0.0f ? 1.0f : -1.0f; }                             /* Mathf.Sign (Q10) */
static inline float mag2(float x, float z) { return sqrtf(x * x + z * z); }                          /* Vector3.magnitude, y = 0 */
static inline float mag3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

static inline int is_straight(const hko_env* e, int section) { return e->sec[section % e->L].track_inside_radius == 0.0f; } /* DPT:195, REC:758 */

static inline float kart_steer(const hko_env* e, float acc_ang_v)
{   /* AK:300 */
    const hk_kart_stats* s = &e->cfg.stats;
    return f_clamp(s->MaxSteer * hk_expf(-acc_ang_v / s->TireWearRate), s->MinSteer, s->MaxSteer);
}
static inline float tire_wear_proportion(const hko_env* e, float steer)
{   /* AK:304-307 */
    const hk_kart_stats* s = &e->cfg.stats;
    return (s->MaxSteer - steer) / (s->MaxSteer - s->MinSteer);
}
static inline float max_lat_gs_for_wear(const hko_env* e, float wear)
{   /* AK:517-520 */
    const hk_kart_stats* s = &e->cfg.stats;
    return (1 - wear) * (s->MaxGs - s->MinGs) + s->MinGs;
}
static inline float turning_radius(float vx, float vz, float fx, float fz, float wy)
{   /* AK:522-529 */
    float out = (vx * fx + vz * fz) / wy;
    if (isinf(out) || isnan(out)) return 1000.0f;
    return out;
}
static float max_speed_for_state(const hko_env* e, const hk_agent_state* a)
{   /* AK:531-547 getMaxSpeedForState -> getMaxSpeedForRadiusAndWear */
    const hk_kart_stats* s = &e->cfg.stats;
    float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
    float radius = turning_radius(a->vx, a->vz, fx, fz, a->wy);
    float wear = tire_wear_proportion(e, a->final_steer);
    if (radius == 0) return s->TopSpeed;
    float allowed = sqrtf(max_lat_gs_for_wear(e, wear) * 9.81f * f_abs(radius));
    if (isinf(allowed) || isnan(allowed)) allowed = s->TopSpeed;
    return f_clamp(allowed, 0.0001f, s->TopSpeed);
}

/* ------------------------------------------------------------------ analytic Physics.Raycast */
/* ray (o, unit d) vs one wall segment: returns t >= 0 (distance) or -1.  With e = p1 - p0, w = p0 - o:
 *   den = d x e,  s = (w x d) / den (position along the segment),  t = (w x e) / den (distance along the ray).
 * The crossing test 0 <= s <= 1, t >= 0 is made on the cross products themselves (signs and |s-numerator| <= |den|), so no
 * rounded quotient is compared; only the distance of an actual hit is divided.  (This is our own restatement of
 * Physics.Raycast against a wall slice — DESIGN.md section 4 — and the kernels use exactly the same expressions.) */
static inline float ray_seg(float ox, float oz, float dx, float dz, const hk_wall_seg* w)
{
    float ex = w->x1 - w->x0, ez = w->z1 - w->z0;
    float den = dx * ez - dz * ex;
    float wx = w->x0 - ox, wz = w->z0 - oz;
    float tn = wx * ez - wz * ex;
    float sn = wx * dz - wz * dx;
    int hit = den > 0.0f ? (sn >= 0.0f && sn <= den && tn >= 0.0f) : (den < 0.0f && sn <= 0.0f && sn >= den && tn <= 0.0f);
    if (!hit) return -1.0f;
    return tn / den;
}

/* Physics.Raycast(origin, dir, maxDistance, TrackMask, Ignore triggers): nearest wall hit within maxdist, else -1 */
float hko_raycast_track(hko_env* e, float ox, float oz, float dx, float dz, float maxdist)
{
    float best = -1.0f;
    for (int i = 0; i < e->NW; i++) {
        float t = ray_seg(ox, oz, dx, dz, &e->walls[i]);
        if (t >= 0.0f && t <= maxdist && (best < 0.0f || t < best)) best = t;
    }
    return best;
}

/* ray vs a kart's capsule sliced at the ray height = stadium (radius r, core segment along the kart's forward).
 * Unity: a ray starting inside a collider does not hit it (Q10). returns distance or -1 */
static float ray_stadium(float ox, float oz, float dx, float dz, const hk_agent_state* k, float r)
{
    float fx = hk_sinf(k->yaw), fz = hk_cosf(k->yaw);
    float rx = fz, rz = -fx;                       /* right = (cos yaw, -sin yaw) */
    float relx = ox - k->px, relz = oz - k->pz;
    float lx = relx * rx + relz * rz;              /* ray origin in kart-local (x right, z forward) */
    float lz = relx * fx + relz * fz;
    float ldx = dx * rx + dz * rz;
    float ldz = dx * fx + dz * fz;
    /* inside test: distance from origin to the core segment <= r */
    float cz = f_clamp(lz, KART_CAP_Z0, KART_CAP_Z1);
    float ddz = lz - cz;
    if (lx * lx + ddz * ddz <= r * r) return -1.0f;
    float best = -1.0f;
    /* rectangle [-r, r] x [z0, z1]: slab entry */
    {
        float tmin = 0.0f, tmax = 3.0e38f;
        int ok = 1;
        if (ldx == 0.0f) { if (lx < -r || lx > r) ok = 0; }
        else {
            float t1 = (-r - lx) / ldx, t2 = (r - lx) / ldx;
            if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
            tmin = f_max(tmin, t1); tmax = f_min(tmax, t2);
        }
        if (ok) {
            if (ldz == 0.0f) { if (lz < KART_CAP_Z0 || lz > KART_CAP_Z1) ok = 0; }
            else {
                float t1 = (KART_CAP_Z0 - lz) / ldz, t2 = (KART_CAP_Z1 - lz) / ldz;
                if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
                tmin = f_max(tmin, t1); tmax = f_min(tmax, t2);
            }
        }
        if (ok && tmin <= tmax) best = tmin;
    }
    /* end circles */
    for (int c = 0; c < 2; c++) {
        float czc = c == 0 ? KART_CAP_Z0 : KART_CAP_Z1;
        float mx = lx, mz = lz - czc;
        float b = mx * ldx + mz * ldz;
        float cc = mx * mx + mz * mz - r * r;
        float disc = b * b - cc;
        if (disc < 0.0f) continue;
        float t = -b - sqrtf(disc);
        if (t >= 0.0f && (best < 0.0f || t < best)) best = t;
    }
    return best;
}

/* Physics.Raycast(..., AgentMask): nearest enabled OTHER kart within maxdist; *who = its index */
static float raycast_agents(const hko_env* e, const hk_agent_state* ags, int self, float ox, float oz, float dx, float dz,
                            float maxdist, int* who)
{
    float best = -1.0f;
    if (who) *who = -1;
    for (int j = 0; j < e->A; j++) {
        if (j == self || !(ags[j].flags & HK_F_ENABLED)) continue;
        float t = ray_stadium(ox, oz, dx, dz, &ags[j], e->ray_agent_r);
        if (t >= 0.0f && t <= maxdist && (best < 0.0f || t < best)) { best = t; if (who) *who = j; }
    }
    return best;
}

static inline void sensor_ray(const hko_env* e, const hk_agent_state* k, int si, float* ox, float* oz, float* dx, float* dz)
{
    float fx = hk_sinf(k->yaw), fz = hk_cosf(k->yaw);
    *ox = k->px + SENSOR_LZ * fx;
    *oz = k->pz + SENSOR_LZ * fz;
    /* Sensor.Transform.forward = parent rotation * local rotation * (0,0,1): the kart's forward turned by the sensor's
     * local yaw (Unity Y rotation: +z toward +x) */
    *dx = fx * e->sens_c[si] + fz * e->sens_s[si];
    *dz = fz * e->sens_c[si] - fx * e->sens_s[si];
}

/* ------------------------------------------------------------------ closest points between 2-D segments */
/* returns squared distance, closest point on segment 1 (p1->q1) in (*c1x,*c1z) and on segment 2 in (*c2x,*c2z) */
static float seg_seg_closest(float p1x, float p1z, float q1x, float q1z, float p2x, float p2z, float q2x, float q2z,
                             float* c1x, float* c1z, float* c2x, float* c2z)
{
    float d1x = q1x - p1x, d1z = q1z - p1z;
    float d2x = q2x - p2x, d2z = q2z - p2z;
    float rx = p1x - p2x, rz = p1z - p2z;
    float a = d1x * d1x + d1z * d1z;
    float ee = d2x * d2x + d2z * d2z;
    float f = d2x * rx + d2z * rz;
    float s, t;
    const float EPS = 1e-12f;
    if (a <= EPS && ee <= EPS) { s = 0.0f; t = 0.0f; }
    else if (a <= EPS) { s = 0.0f; t = f_clamp(f / ee, 0.0f, 1.0f); }
    else {
        float c = d1x * rx + d1z * rz;
        if (ee <= EPS) { t = 0.0f; s = f_clamp(-c / a, 0.0f, 1.0f); }
        else {
            float b = d1x * d2x + d1z * d2z;
            float den = a * ee - b * b;
            if (den != 0.0f) s = f_clamp((b * f - c * ee) / den, 0.0f, 1.0f); else s = 0.0f;
            t = (b * s + f) / ee;
            if (t < 0.0f) { t = 0.0f; s = f_clamp(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = f_clamp((b - c) / a, 0.0f, 1.0f); }
        }
    }
    *c1x = p1x + d1x * s; *c1z = p1z + d1z * s;
    *c2x = p2x + d2x * t; *c2z = p2z + d2z * t;
    float ddx = *c1x - *c2x, ddz = *c1z - *c2z;
    return ddx * ddx + ddz * ddz;
}

static inline void kart_core(const hk_agent_state* k, float px, float pz, float* ax, float* az, float* bx, float* bz)
{
    float fx = hk_sinf(k->yaw), fz = hk_cosf(k->yaw);
    *ax = px + KART_CAP_Z0 * fx; *az = pz + KART_CAP_Z0 * fz;
    *bx = px + KART_CAP_Z1 * fx; *bz = pz + KART_CAP_Z1 * fz;
}

/* ------------------------------------------------------------------ plans */
/* HKA.planFixed :145-166 */
static void plan_fixed(const hko_env* e, int agent, hk_agent_state* a)
{
    int depth = e->cfg.tree_search_depth[agent];
    int hi = a->section_index + depth; if (hi > 1000) hi = 1000;
    for (int i = a->section_index + 1; i < hi + 1; i++) {
        int key = i % e->L;
        if (a->plan_lane[key] == 0) {
            int lane = e->sec[(i - 1) % e->L].optimal_lane;          /* getOptimalNextLane DPT:250 */
            a->plan_lane[key] = (uint8_t)lane;
            a->plan_vel[key] = e->max_speed;                          /* m_Kart.GetMaxSpeed() */
        }
    }
}

/* draws of the Training-mode code (stand-ins for UnityEngine.Random / System.Random / MathNet Normal, see hk.h) */
typedef struct { uint32_t k0, k1, c1, c2, c3, n; } tr_rng;
static uint32_t tr_u32(tr_rng* g) { uint32_t r[4]; philox4x32(g->n++, g->c1, g->c2, g->c3, g->k0, g->k1, r); return r[0]; }
static int tr_range_i(tr_rng* g, int lo, int hi)        /* Random.Range(int, int): [lo, hi) */
{
    if (hi <= lo) return lo;
    return lo + (int)(((uint64_t)tr_u32(g) * (uint64_t)(hi - lo)) >> 32);
}
static float tr_range_f(tr_rng* g, float lo, float hi) { return lo + u01(tr_u32(g)) * (hi - lo); }   /* Random.Range(float, float) */
static float tr_normal(tr_rng* g)
{
    uint32_t r[4]; philox4x32(g->n++, g->c1, g->c2, g->c3, g->k0, g->k1, r);
    float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(r[1]);
    return sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);
}
static float tr_gauss_bounded(tr_rng* g, float mean, float sd, float lo, float hi)
{   /* KartMCTS.NextGaussian(mean, sd, min, max) KM:225-240 */
    float x; int attempts = 0;
    do { x = mean + tr_normal(g) * sd; attempts += 1; } while ((x < lo || x > hi) && attempts < 10);
    if (attempts == 10 && (x < lo || x > hi)) return mean;
    return x;
}

/* HKA.planRandomly :109-143 */
static void plan_randomly(const hko_env* e, int env, int agent, hk_agent_state* a)
{
    const hk_env_state* es = &e->es[env];
    tr_rng g = {e->cfg.train_seed ^ 0x504C414Eu, (uint32_t)(e->cfg.env_id_base + env) * (uint32_t)e->A + (uint32_t)agent,
                (uint32_t)es->episode_steps, (uint32_t)es->episodes_done, 0u, 0u};
    int depth = e->cfg.tree_search_depth[agent];
    int hi = a->section_index + depth; if (hi > 1000) hi = 1000;
    for (int i = a->section_index + 1; i < hi + 1; i++) {
        int key = i % e->L;
        if (a->plan_lane[key] != 0) continue;
        int index = (int)__builtin_rintf(f_abs(tr_gauss_bounded(&g, 0.0f, 1.0f, -(float)4 + 1.0f, (float)4 - 1.0f)));
        int ol = e->sec[(i - 1) % e->L].optimal_lane;
        int sign = ol == 1 ? 1 : (ol == 4 ? -1 : 0);                 /* getOptimalLaneSign DPT:221-231 */
        int lane = sign < 0 ? 4 - index : 1 + index;                 /* Enumerable.Range(1, 4).OrderBy(l => sign * l)[index] */
        a->plan_lane[key] = (uint8_t)lane;
        if (e->cfg.high_mode[agent] == HK_HIGH_FIXED) a->plan_vel[key] = e->max_speed;
        else a->plan_vel[key] = e->max_speed - f_abs(tr_gauss_bounded(&g, 0.0f, 1.5f, -8.0f, 8.0f));
    }
}

/* REC.ResetGame :520-668, Training mode: where every kart starts.  out: section, lane, genTWP, distFromSpawn per agent */
static void training_layout(hko_env* e, int env, const int* ord, int* sec, int* lane, float* twp, float* dist)
{
    const hk_env_state* es = &e->es[env];
    tr_rng g = {e->cfg.train_seed, (uint32_t)(e->cfg.env_id_base + env), (uint32_t)es->episodes_done, (uint32_t)es->experiment_num, 0x54524E47u, 0u};
    const int A = e->A, L = e->L, goal = e->cfg.laps * L + 1;
    const int headToHead = tr_range_i(&g, 0, 9) >= 3;                                   /* :523 */
    int used_sec[HK_MAX_AGENTS], used_lane[HK_MAX_AGENTS], n_added = 0, initialSection = -1;
    for (int j = 0; j < A; j++) {
        const int i = ord[j];
        int s_i, l_i;
        if (!headToHead) {                                                              /* :533-580 */
            while (1) {
                s_i = tr_range_i(&g, 0, goal);
                l_i = tr_range_i(&g, 1, 5);
                int clash = 0;
                for (int q = 0; q < n_added; q++) clash |= (used_sec[q] == s_i % L && used_lane[q] == l_i);
                if (!clash) break;
            }
            twp[i] = tr_range_f(&g, 0.0f, 1.0f);
        } else if (n_added == 0) {                                                      /* :585-629 */
            s_i = tr_range_i(&g, 0, goal);
            initialSection = s_i;
            twp[i] = tr_range_f(&g, 0.0f, 1.0f);
            l_i = tr_range_i(&g, 1, 5);
        } else {                                                                        /* :631-684 */
            const int lo = initialSection - 1 > 0 ? initialSection - 1 : 0, hi = initialSection + 2 < goal ? initialSection + 2 : goal;
            while (1) {
                s_i = tr_range_i(&g, lo, hi);
                l_i = tr_range_i(&g, 1, 5);
                int clash = 0;
                for (int q = 0; q < n_added; q++) clash |= (used_sec[q] == s_i % L && used_lane[q] == l_i);
                if (!clash) break;
            }
            twp[i] = tr_range_f(&g, 0.0f, 1.0f);
        }
        float d = tr_range_f(&g, 1.0f, 4.0f);
        if (tr_range_f(&g, 0.0f, 1.0f) < 0.3f) {                                        /* start close behind a wall */
            const hk_section* s = &e->sec[s_i % L];
            const sec_pre* sp = &e->sp[s_i % L];
            float hit = hko_raycast_track(e, s->lane_x[l_i - 1], s->lane_z[l_i - 1], sp->fx, sp->fz, 10.0f);
            if (hit >= 0.0f) d = hit - 1.0f;
        }
        sec[i] = s_i; lane[i] = l_i; dist[i] = d;
        used_sec[n_added] = s_i % L; used_lane[n_added] = l_i; n_added++;
    }
}

/* KA.SetZeroInputs :480-486, KA.Deactivate :405-416 */
static void deactivate(const hko_env* e, hk_agent_state* a)
{
    a->steering = 0.0f;
    a->flags &= ~(HK_F_ACCEL | HK_F_BRAKE);
    a->vx = 0.0f; a->vz = 0.0f; a->wy = 0.0f;
    a->flags &= ~HK_F_CAN_MOVE;
    if (e->cfg.disable_on_end) a->flags &= ~HK_F_ENABLED;
    a->flags &= ~HK_F_ACTIVE;
}

/* ------------------------------------------------------------------ REC.ResetGame :499-719 (Experiment / Race grid) */
static void reset_env(hko_env* e, int env)
{
    hk_env_state* es = &e->es[env];
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    /* expLaneChoices {2, 3, 2, 3} / expSectionChoices {0, 0, 1, 1} (:526-527) cover the 4 agents of the largest reference scene
       (a fifth agent would index past them).  For the synthetic 8-agent configuration the same pattern continues row by row:
       grid slot j starts in section j / 2, lane 2 + j % 2 — identical to the reference's tables for j < 4. */
    es->episode_steps = 0;                                   /* :505 */
    es->inactive_mask = 0;                                   /* :506 */
    if (e->sec_min_time) hko_rw_reset_env(e, env);           /* :508-512 */
    const int* ord = &e->perms[(size_t)(((es->experiment_num % e->nperm) + e->nperm) % e->nperm) * e->A];  /* :528 */
    const uint32_t env_gid = (uint32_t)(e->cfg.env_id_base + env);
    const int training = e->cfg.env_mode == HK_MODE_TRAINING;
    int t_sec[HK_MAX_AGENTS], t_lane[HK_MAX_AGENTS];
    float t_twp[HK_MAX_AGENTS], t_dist[HK_MAX_AGENTS];
    if (training) training_layout(e, env, ord, t_sec, t_lane, t_twp, t_dist);
    float old_steer[HK_MAX_AGENTS];                          /* m_FinalStats.Steer as the previous episode left it */
    for (int i = 0; i < e->A; i++) old_steer[i] = ags[i].final_steer;
    for (int j = 0; j < e->A; j++) {
        int i = ord[j];
        hk_agent_state* a = &ags[i];
        const int t_laps = a->tele_completed_laps, t_step = a->tele_lap_end_step;
        const float t_last = a->tele_last_lap, t_best = a->tele_best_lap, t_total = a->tele_total_time;
        memset(a, 0, sizeof(*a));
        /* the TelemetryViewer is not reset with the game: its arrays survive (it notices the lower lap count itself) */
        a->tele_completed_laps = t_laps; a->tele_lap_end_step = t_step;
        a->tele_last_lap = t_last; a->tele_best_lap = t_best; a->tele_total_time = t_total;
        a->section_index = training ? t_sec[i] : (j >> 1);   /* :583 / :634 */
        a->init_checkpoint_index = a->section_index;         /* :586 */
        a->acc_ang_v = e->init_acc_ang_v;                    /* :588 */
        if (training) {
            const hk_kart_stats* st = &e->cfg.stats;
            a->acc_ang_v = -st->TireWearRate * hk_logf(1 - ((st->MaxSteer - st->MinSteer) * t_twp[i] / st->MaxSteer));
        }
        a->lane = training ? t_lane[i] : 2 + (j & 1);              /* :593 */
        const float spawn = training ? t_dist[i] : 3.0f;
        const hk_section* s = &e->sec[a->section_index % e->L];
        const sec_pre* sp = &e->sp[a->section_index % e->L];
        float yaw = sp->yaw_rad;                             /* :602 rotation = lane marker rotation */
        float px = s->lane_x[a->lane - 1] + sp->fx * spawn;  /* :614 position + forward * distFromSpawn (3.0 outside Training) */
        float pz = s->lane_z[a->lane - 1] + sp->fz * spawn;
        if (e->cfg.jitter_seed != 0u) {                      /* synthetic, NOT in the reference */
            uint32_t r[4];
            philox4x32((uint32_t)es->experiment_num, (uint32_t)i, 0u, 0u, e->cfg.jitter_seed + env_gid, 0u, r);
            px += (2.0f * u01(r[0]) - 1.0f) * e->cfg.jitter_pos;
            pz += (2.0f * u01(r[1]) - 1.0f) * e->cfg.jitter_pos;
            yaw += (2.0f * u01(r[2]) - 1.0f) * e->cfg.jitter_yaw;
            if (yaw < 0.0f) yaw += TWO_PI_F;
            if (yaw >= TWO_PI_F) yaw -= TWO_PI_F;
        }
        a->px = px; a->pz = pz; a->yaw = yaw;
        /* prepareForReuse KA:209-221 (UpdateStats, counters, plans cleared) + initialPlan HKA:84-104 */
        a->final_steer = kart_steer(e, a->acc_ang_v);
        if (e->cfg.training_agent[i]) plan_randomly(e, env, i, a);                      /* Mode == Training: HKA:101-103 */
        else if (e->cfg.high_mode[i] == HK_HIGH_FIXED) plan_fixed(e, i, a);
        /* Activate KA:421-435 */
        a->flags = HK_F_ACTIVE | HK_F_ENABLED;               /* m_CanMove stays false until StartRaceAfterDelay */
    }
    if (e->mcts) {
        /* the sectionTimes back-fill of REC:679-702, then agent by agent prepareForReuse HKA:428-452 (beliefs, bestStates,
         * currentRoot cleared; the search thread aborted) and initialPlan -> planWithMCTS(T: 1.5) HKA:84-96 (REC:705-710).  The
         * interleaving is observable in one value: prepareForReuse is what refreshes m_FinalStats.Steer (UpdateStats KA:213), so
         * agent i's first plan reads the tire age of every agent j > i from the Steer its last tick of the PREVIOUS episode left
         * (HKA:236) — `old_steer`.  (sectionTimes[own section] = 0, KA:212, was already set by the back-fill.) */
        for (int i = 0; i < e->A; i++) {
            hk_mcts_state* m = &e->mcts[(size_t)env * e->A + i];
            const int searches = m->searches;
            memset(m, 0, sizeof(*m));
            m->searches = searches;
            m->ready_step = -1;
            hko_mcts_tree_drop(e, env, i);
        }
        hko_mcts_backfill_section_times(e, env);
        for (int i = 0; i < e->A; i++) {
            if (e->cfg.high_mode[i] != HK_HIGH_MCTS || e->cfg.training_agent[i]) continue;
            float seen[HK_MAX_AGENTS];
            for (int j = 0; j < e->A; j++) seen[j] = j > i ? old_steer[j] : ags[j].final_steer;
            hko_mcts_request(e, env, i, e->cfg.mcts_initial_iterations, e->cfg.mcts_initial_latency_ticks, seen);
        }
    }
}

/* REC.ResetGame :679-702 — "Generate times for reaching certain sections": every kart gets a made-up history of section times
 * from the rearmost kart's section up to its own, each drawn below the one before (int Random.Range(earliest, 0)), and 0 for the
 * section it stands in.  planWithMCTS reads them as the time offsets of karts behind the furthest one (HKA:221-224).
 * UnityEngine.Random -> Philox keyed by mcts_seed, the global env id, the agent, the episode and the section ("parity unpinned").
 * The same loop also counts agentsPastSection[team][section]; those counts are dead: ApplySectionRewardsAndPenalties reads
 * agentsPastSection[t][s] only where minSectionTimes[t] holds s, and the statement that creates that key assigns the count
 * (REC:366-370), so a back-filled count is always overwritten before it is read.  Not restated. */
void hko_mcts_backfill_section_times(hko_env* e, int env)
{
    const hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    const uint32_t env_gid = (uint32_t)(e->cfg.env_id_base + env);
    int back = ags[0].section_index;
    for (int i = 1; i < e->A; i++) if (ags[i].section_index < back) back = ags[i].section_index;   /* furthestBackSection :546,591 */
    for (int i = 0; i < e->A; i++) {
        hk_mcts_state* m = &e->mcts[(size_t)env * e->A + i];
        int earliest = -e->cfg.max_episode_steps;                                                   /* :687 */
        for (int tp = back; tp < ags[i].section_index; tp++) {
            uint32_t r[4];
            philox4x32((uint32_t)tp, (uint32_t)i, (uint32_t)e->es[env].episodes_done, 0x53454354u, e->cfg.mcts_seed, env_gid, r);
            const int v = earliest + (int)(((uint64_t)r[0] * (uint64_t)(uint32_t)(-earliest)) >> 32);   /* [earliest, 0) */
            m->sec_time[tp & (HK_MCTS_SECTIME_RING - 1)] = v;
            earliest = v;
        }
        m->sec_time[ags[i].section_index & (HK_MCTS_SECTIME_RING - 1)] = 0;                          /* :697 */
    }
}

static void snapshot_results(hko_env* e, int env)
{
    hk_env_state* es = &e->es[env];
    for (int i = 0; i < e->A; i++) {
        const hk_agent_state* a = &e->ag[(size_t)env * e->A + i];
        hk_episode_result* r = &e->res[(size_t)env * e->A + i];
        r->time_steps = a->time_steps;
        r->section_index = a->section_index;
        r->illegal_lane_changes = a->illegal_lane_changes;
        r->forward_collisions = a->forward_collisions;
        r->avg_lane_diff = a->avg_lane_diff;
        r->avg_vel_diff = a->avg_vel_diff;
        r->reward = a->cum_reward;
        r->episode = es->episodes_done;
        r->last_lap = a->tele_last_lap; r->best_lap = a->tele_best_lap; r->total_time = a->tele_total_time;
        r->laps_completed = a->tele_completed_laps; r->lap_end_step = a->tele_lap_end_step;
        r->speed = mag3(a->vx, 0.0f, a->vz);
        r->active = (a->flags & HK_F_ACTIVE) ? 1 : 0;
        r->group_reward = a->group_reward;
    }
}

/* ------------------------------------------------------------------ HKA.SolveLQR :699-1236 */
typedef struct { float x, z; } pt2;

static inline pt2 lane_marker(const hko_env* e, int idx, int lane)
{   /* DPT.getBoxColliderForLane :99-111 (lane 0 / invalid -> Trigger) */
    pt2 p;
    const hk_section* s = &e->sec[idx];
    if (lane >= 1 && lane <= 4) { p.x = s->lane_x[lane - 1]; p.z = s->lane_z[lane - 1]; }
    else { p.x = s->trig_x; p.z = s->trig_z; }
    return p;
}

/* HKA:1341-1344 */
static inline double angle_difference(double a1, double a2) { return hk_atan2(hk_sin(a2 - a1), hk_cos(a2 - a1)); }

/* BoxCollider.ClosestPoint(p) distance for the section Trigger (HKA:846,876), evaluated in the box frame */
static float dist_to_trigger_box(const hko_env* e, int idx, float px, float pz)
{
    const hk_section* s = &e->sec[idx];
    const sec_pre* sp = &e->sp[idx];
    float relx = px - s->trig_x, relz = pz - s->trig_z;
    float lx = relx * sp->fz + relz * (-sp->fx);   /* right = (cos, -sin) = (fz, -fx) */
    float lz = relx * sp->fx + relz * sp->fz;
    float dx = lx - f_clamp(lx, -TRIG_HX, TRIG_HX);
    float dz = lz - f_clamp(lz, -TRIG_HZ, TRIG_HZ);
    return sqrtf(dx * dx + dz * dz);
}

static void solve_lqr(hko_env* e, int env, int ego)
{
    const hk_config* cfg = &e->cfg;
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    hk_agent_state* me = &ags[ego];
    const int A = e->A, L = e->L;
    const int fixed = cfg->high_mode[ego] == HK_HIGH_FIXED;
    const float dy = e->sec[0].marker_y - cfg->kart_y;      /* Q13: constant y offset kart <-> markers */
    /* :702 allPlayers = [this] + teamAgents + otherAgents */
    int all[HK_MAX_AGENTS], nall = 0;
    all[nall++] = ego;
    for (int j = 0; j < cfg->n_team[ego]; j++) all[nall++] = cfg->team_agents[ego][j];
    for (int j = 0; j < cfg->n_other[ego]; j++) all[nall++] = cfg->other_agents[ego][j];
    int pl[HK_MAX_AGENTS], N = 0;
    int nearbyAgents = -1;                                   /* :708 */
    if (A > 2) {                                             /* :709-720 */
        for (int i = 0; i < nall; i++) {
            const hk_agent_state* k = &ags[all[i]];
            if (mag3(k->px - me->px, 0.0f, k->pz - me->pz) < 8) { nearbyAgents += 1; pl[N++] = all[i]; }
        }
    } else {
        for (int i = 0; i < nall; i++) pl[N++] = all[i];     /* :723 */
    }
    nearbyAgents = nearbyAgents > 1 ? nearbyAgents : 1;      /* :725 */
    const double dt = (double)cfg->dt;                       /* :707 Time.fixedDeltaTime (float) widened */
    const int n = 4 * N;
    double Am[HKO_MAX_PLAYERS * 16], Bm[HKO_MAX_PLAYERS * 8];
    static __thread double Qm[HKO_MAX_PLAYERS * HKO_MAX_N * HKO_MAX_N];
    double qm[HKO_MAX_PLAYERS * HKO_MAX_N], Rm[HKO_MAX_PLAYERS * 4], x0[HKO_MAX_N];
    hk_lq_debug* dbg = e->dbg ? &e->dbg[(size_t)env * A + ego] : NULL;
    if (dbg) { memset(dbg, 0, sizeof(*dbg)); dbg->n_players = N; }

    for (int i = 0; i < N; i++) {                            /* :726 */
        const int ki = pl[i];
        const hk_agent_state* k = &ags[ki];
        const float kfx = hk_sinf(k->yaw), kfz = hk_cosf(k->yaw);   /* transform.forward */
        const float speed = mag3(k->vx, 0.0f, k->vz);               /* Rigidbody.velocity.magnitude */
        double initial[4];
        initial[XI] = k->px; initial[ZI] = k->pz; initial[VI] = speed;     /* :731-733 */
        float heading = hk_atan2f(kfz, kfx);                               /* :734 */
        if (heading < 0) heading += TWO_PI_F;                              /* :735 */
        initial[HI] = heading;
        hko_bicycle_AB(dt, initial, Am + i * 16, Bm + i * 8);              /* :739 */
        for (int c = 0; c < 4; c++) x0[4 * i + c] = initial[c];            /* :742 */
        double target[4];
        int s = k->section_index + 1;                                      /* :746 */
        int idx = s % L;
        /* :750-777 target lane / velocity: own plan, or the ego's BELIEF about k's plan (opponentUpcomingLanes,
         * only ever filled by the MCTS planner, HKA:395-400) -> empty for Fixed agents */
        pt2 lane; double vel;
        const hk_mcts_state* bel = e->mcts ? &e->mcts[(size_t)env * A + ego] : NULL;   /* the EGO's beliefs (HKA:77-78) */
        if (ki == ego && me->plan_lane[idx] != 0) {
            lane = lane_marker(e, idx, me->plan_lane[idx]);
            double pv = me->plan_vel[idx] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
            vel = (double)e->max_speed < pv ? (double)e->max_speed : pv;   /* Math.Min :757 */
        } else if (ki != ego && bel && bel->belief_lane[ki][idx] != 0) {   /* :767-771 */
            lane = lane_marker(e, idx, bel->belief_lane[ki][idx]);
            double pv = (float)bel->belief_vel[ki][idx] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
            vel = (double)e->max_speed < pv ? (double)e->max_speed : pv;
        } else {
            lane = lane_marker(e, idx, 0); vel = e->max_speed;             /* :761-762 / :774-775 */
        }
        pt2 center = lane_marker(e, idx, 0);                               /* :778 centerLine */
        int idx2 = (s + 1) % L;                                            /* :781 */
        pt2 nextLane; double nextVel;
        if (ki == ego && me->plan_lane[idx2] != 0) {
            nextLane = lane_marker(e, idx2, me->plan_lane[idx2]);
            double pv = me->plan_vel[idx2] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
            nextVel = (double)e->max_speed < pv ? (double)e->max_speed : pv;
        } else if (ki != ego && bel && bel->belief_lane[ki][idx2] != 0) {  /* :797-801 */
            nextLane = lane_marker(e, idx2, bel->belief_lane[ki][idx2]);
            double pv = (float)bel->belief_vel[ki][idx2] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
            nextVel = (double)e->max_speed < pv ? (double)e->max_speed : pv;
        } else {
            nextLane = lane_marker(e, idx2, 0); nextVel = e->max_speed;
        }
        target[XI] = lane.x; target[ZI] = lane.z;                          /* :808-809 */
        target[VI] = (speed <= 5.0f) ? 0.0f : vel;                         /* :810-817 */
        /* :819-926 heading heuristic */
        double finalTargetHeading;
        int branch;
        float targetHeading = hk_atan2f(lane.z - k->pz, lane.x - k->px);   /* :821 */
        if (targetHeading < 0) targetHeading += TWO_PI_F;
        const int kstraight = is_straight(e, k->section_index);
        if (mag3(lane.x - k->px, dy, lane.z - k->pz) <= (kstraight ? 10.5f : 7.5f)) {   /* :823 */
            float h1 = hk_atan2f(lane.z - k->pz, lane.x - k->px);          /* :826 */
            float h2 = hk_atan2f(nextLane.z - lane.z, nextLane.x - lane.x);/* :827 */
            float h5 = hk_atan2f(center.z - k->pz, center.x - k->px);      /* :830 */
            float h6 = hk_atan2f(nextLane.z - k->pz, nextLane.x - k->px);  /* :831 */
            /* :832 cutTrack: ray lane -> nextLane, length |lane - nextLane| */
            float cdx = nextLane.x - lane.x, cdz = nextLane.z - lane.z;
            float clen = mag3(lane.x - nextLane.x, 0.0f, lane.z - nextLane.z);
            int cutTrack = 0;
            if (clen > 0.0f) cutTrack = hko_raycast_track(e, lane.x, lane.z, cdx / clen, cdz / clen, clen) >= 0.0f;
            float ox, oz, rdx, rdz;
            sensor_ray(e, k, 0, &ox, &oz, &rdx, &rdz);
            int hit0 = hko_raycast_track(e, ox, oz, rdx, rdz, speed * 0.5f) >= 0.0f;    /* :834 */
            sensor_ray(e, k, 2, &ox, &oz, &rdx, &rdz);
            int hit1 = hko_raycast_track(e, ox, oz, rdx, rdz, 2.0f) >= 0.0f;            /* :837 */
            sensor_ray(e, k, 4, &ox, &oz, &rdx, &rdz);
            int hit2 = hko_raycast_track(e, ox, oz, rdx, rdz, 1.5f) >= 0.0f;            /* :839 */
            sensor_ray(e, k, 8, &ox, &oz, &rdx, &rdz);
            int hit3 = hko_raycast_track(e, ox, oz, rdx, rdz, 1.5f) >= 0.0f;            /* :841 */
            sensor_ray(e, k, 6, &ox, &oz, &rdx, &rdz);
            int hit4 = hko_raycast_track(e, ox, oz, rdx, rdz, 2.0f) >= 0.0f;            /* :843 */
            float dC = dist_to_trigger_box(e, idx, k->px, k->pz);
            int side = hit1 || hit2 || hit3 || hit4;
            if (cutTrack && dC > 4.0f) {                                                /* :846 B1 */
                branch = 1;
                if (h5 < 0) h5 += TWO_PI_F;
                finalTargetHeading = h5;
                if (finalTargetHeading < 0) finalTargetHeading += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], finalTargetHeading);
            } else if ((side && (f_sign(h1) == f_sign(h5))) || hit0) {                  /* :857 B2 (Q12) */
                branch = 2;
                if (h5 < 0) h5 += TWO_PI_F;
                finalTargetHeading = h5 - angle_difference(h1, h5) * 0.7f;              /* :860 */
                if (finalTargetHeading < 0) finalTargetHeading += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], finalTargetHeading);
            } else if (side && (f_sign(h1) != f_sign(h5))) {                            /* :867 B3 */
                branch = 3;
                if (h5 < 0) h5 += TWO_PI_F;
                finalTargetHeading = h5;
                if (finalTargetHeading < 0) finalTargetHeading += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], finalTargetHeading);
            } else if (dC <= 4.0f) {                                                    /* :876 B4 */
                branch = 4;
                target[XI] = nextLane.x; target[ZI] = nextLane.z;
                if (speed > 5.0f) target[VI] = nextVel;
                if (h6 < 0) h6 += TWO_PI_F;
                finalTargetHeading = h6;
                if (finalTargetHeading < 0) finalTargetHeading += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], finalTargetHeading);
            } else {                                                                    /* :891 B5 */
                branch = 5;
                if (h1 < 0) h1 += TWO_PI_F;
                if (h2 < 0) h2 += TWO_PI_F;
                finalTargetHeading = h1 - angle_difference(h2, h1) * 0.4f;              /* :898 */
                if (finalTargetHeading < 0) finalTargetHeading += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], finalTargetHeading);
            }
        } else {
            float ox, oz, rdx, rdz;
            sensor_ray(e, k, 0, &ox, &oz, &rdx, &rdz);
            int hit = hko_raycast_track(e, ox, oz, rdx, rdz, kstraight ? 8.0f : 5.0f) >= 0.0f;   /* :906 */
            if (hit) {                                                                  /* B6 */
                branch = 6;
                float h1 = hk_atan2f(center.z - k->pz, center.x - k->px);               /* :912 */
                if (h1 < 0) h1 += TWO_PI_F;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], h1) * 0.85f;   /* :915 */
            } else {                                                                    /* B7 */
                branch = 7;
                finalTargetHeading = initial[HI] - angle_difference(initial[HI], targetHeading);/* :921 */
            }
        }
        target[HI] = finalTargetHeading;                                                /* :926 */
        /* :930-964 target weights */
        double tw[4];
        if (N > 2) tw[HI] = (fixed ? 2.5 : 3.5) * nearbyAgents; else tw[HI] = (fixed ? 1.9 : 3.5);
        if (speed <= 5.0f) {
            tw[XI] = nearbyAgents * 0.3 * 3.1; tw[ZI] = nearbyAgents * 0.3 * 3.1; tw[VI] = nearbyAgents * -2;
        } else {
            double mx = initial[VI] > 1 ? initial[VI] : 1;                              /* Math.Max(1, v) */
            tw[XI] = nearbyAgents * 0.3 * 3.1 / mx; tw[ZI] = nearbyAgents * 0.3 * 3.1 / mx; tw[VI] = nearbyAgents * 5e-4;
        }
        /* :976-1003 multiplier */
        float multiplier;
        if (A > 2 && N > 2) multiplier = (ki == ego ? (fixed ? 0.55f : 1.0f) : 1.7f) / nearbyAgents;
        else multiplier = (ki == ego ? (fixed ? 0.45f : 1.0f) : 1.3f);
        /* :1004-1190 k's opponents then teammates, in k's own order (Q3) */
        double avoid_w[2 * HK_MAX_AGENTS], opp_t[4 * HK_MAX_AGENTS], opp_w[3 * HK_MAX_AGENTS];
        int M = 0, nearbyOpponents = 0;
        int olist[HK_MAX_AGENTS], oteam[HK_MAX_AGENTS], no = 0;
        for (int j = 0; j < cfg->n_other[ki]; j++) { olist[no] = cfg->other_agents[ki][j]; oteam[no++] = 0; }
        for (int j = 0; j < cfg->n_team[ki]; j++) { olist[no] = cfg->team_agents[ki][j]; oteam[no++] = 1; }
        double aw_x[HK_MAX_AGENTS], aw_z[HK_MAX_AGENTS];
        for (int j = 0; j < no; j++) {
            const int oi = olist[j];
            int member = 0;
            for (int q = 0; q < N; q++) if (pl[q] == oi) member = 1;
            if (!member) continue;                                                      /* :1007 / :1101 */
            const hk_agent_state* o = &ags[oi];
            const float dist = mag3(o->px - k->px, 0.0f, o->pz - k->pz);
            const int far = (dist > 8) || !(o->flags & HK_F_ACTIVE);                    /* :1010 / :1104 */
            if (far) { aw_x[M] = 0.0; aw_z[M] = 0.0; }
            else {
                float mult = oteam[j] ? multiplier / 2.0f : multiplier;                 /* :1113 */
                /* 1f / (Mathf.Pow(d, 1.5f) * multiplier): Pow = d*sqrt(d) in double, rounded to float (hk_detmath.h) */
                float pw = (float)((double)dist * sqrt((double)dist));
                float w = 1.0f / (pw * mult);                                           /* :1019 / :1114 */
                aw_x[M] = w; aw_z[M] = w;
                if (!oteam[j]) nearbyOpponents += 1;                                    /* :1023 */
            }
            /* opponent target state :1036-1068 / :1132-1164 */
            int so = o->section_index + 1;
            int io = so % L;
            pt2 ol; double ov;
            if (oi == ego) {
                ol = me->plan_lane[io] != 0 ? lane_marker(e, io, me->plan_lane[io]) : lane_marker(e, io, 0);
                if (oteam[j]) ov = max_speed_for_state(e, me);                          /* :1140,1145 */
                else if (me->plan_lane[io] != 0) {
                    double pv = me->plan_vel[io] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
                    ov = (double)e->max_speed < pv ? (double)e->max_speed : pv;         /* :1044 */
                } else ov = e->max_speed;                                               /* :1049 */
            } else {
                const int bl = bel ? bel->belief_lane[oi][io] : 0;                      /* :1054 / :1150 */
                ol = lane_marker(e, io, bl);
                if (oteam[j]) ov = max_speed_for_state(e, o);                           /* :1153,1158 */
                else if (bl != 0) {
                    double pv = (float)bel->belief_vel[oi][io] + (fixed ? 0 : cfg->velocity_bucket_size[ego] * 2);
                    ov = (double)e->max_speed < pv ? (double)e->max_speed : pv;         /* :1057 */
                } else ov = e->max_speed;                                               /* :1062 */
            }
            opp_t[M * 4 + XI] = ol.x; opp_t[M * 4 + ZI] = ol.z; opp_t[M * 4 + VI] = ov; opp_t[M * 4 + HI] = 0.0;
            /* opponent-target weights :1071-1094 / :1167-1189 */
            double mx = initial[VI] > 1 ? initial[VI] : 1;
            if (!oteam[j]) {
                if (far) { opp_w[M * 3 + 0] = 0.0; opp_w[M * 3 + 1] = 0.0; opp_w[M * 3 + 2] = 0; }
                else if (N > 2) {
                    opp_w[M * 3 + 0] = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents);
                    opp_w[M * 3 + 1] = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents);
                    opp_w[M * 3 + 2] = 0.08 / nearbyAgents;
                } else {
                    opp_w[M * 3 + 0] = (fixed ? 0.1 : 0.2) / mx;
                    opp_w[M * 3 + 1] = (fixed ? 0.1 : 0.2) / mx;
                    opp_w[M * 3 + 2] = 0.08;
                }
            } else {
                if (far || nearbyOpponents < 1) { opp_w[M * 3 + 0] = 0.0; opp_w[M * 3 + 1] = 0.0; opp_w[M * 3 + 2] = 0; }
                else if (N > 2) {
                    opp_w[M * 3 + 0] = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents);
                    opp_w[M * 3 + 1] = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents);
                    opp_w[M * 3 + 2] = 0 / nearbyAgents;
                } else {
                    opp_w[M * 3 + 0] = -(fixed ? 1e-4 : 2e-4) / mx;
                    opp_w[M * 3 + 1] = -(fixed ? 1e-4 : 2e-4) / mx;
                    opp_w[M * 3 + 2] = 0;
                }
            }
            M++;
        }
        for (int j = 0; j < M; j++) { avoid_w[j] = aw_x[j]; avoid_w[M + j] = aw_z[j]; }
        double controlcost = 0.115;                                                     /* :1192 */
        if (N > 2) controlcost = fixed ? 0.135 : 0.25;                                  /* :1193-1196 */
        /* the cost of player i is laid out in ITS OWN order [k, k.others, k.team] and used unpermuted (Q3);
         * its dimension is 4 + 4*M which equals n because every other player is in exactly one of k's lists */
        hko_cost_build(M, target, tw, controlcost, avoid_w, opp_t, opp_w, Qm + (size_t)i * n * n, qm + (size_t)i * n, Rm + i * 4);
        if (dbg) {
            dbg->player_agent[i] = ki; dbg->branch[i] = branch; dbg->control_w[i] = controlcost;
            for (int c = 0; c < 4; c++) { dbg->initial[i][c] = initial[c]; dbg->target[i][c] = target[c]; dbg->target_w[i][c] = tw[c]; }
        }
    }
    double u[2] = {0, 0};
    hko_lq_solve(N, Am, Bm, Qm, qm, Rm, x0, 3, u, NULL);                                 /* :1201 */
    if (dbg) { dbg->u0[0] = u[0]; dbg->u0[1] = u[1]; }
    /* :1206-1224 decode */
    const float maxAng = me->final_steer * 0.4f;                                        /* getMaxAngularVelocity AK:505 */
    float angVel = f_clamp((float)u[1], -maxAng, maxAng);
    if (u[0] < 0) { me->flags &= ~HK_F_ACCEL; me->flags |= HK_F_BRAKE; }
    else if (u[0] > 0) { me->flags |= HK_F_ACCEL; me->flags &= ~HK_F_BRAKE; }
    else { me->flags &= ~(HK_F_ACCEL | HK_F_BRAKE); angVel = 0.0f; }                    /* Q7 */
    me->steering = angVel / (0.4f * me->final_steer);                                   /* :1224 */
}

/* ------------------------------------------------------------------ AK.FixedUpdate / MoveVehicle :243-503 */
static inline void rot_y(float ang_rad, float* x, float* z)
{   /* Quaternion.AngleAxis(deg, up) * v, restated as a planar rotation (Unity Y rotation: +z toward +x) */
    float c = hk_cosf(ang_rad), s = hk_sinf(ang_rad);
    float nx = *x * c + *z * s;
    float nz = *z * c - *x * s;
    *x = nx; *z = nz;
}

static void arcade_kart_update(const hko_env* e, hk_agent_state* a, int rl_agent, float act_steer, int act_branch)
{
    const hk_kart_stats* st = &e->cfg.stats;
    const float dt = e->cfg.dt;
    /* GatherInputs AK:281-293 -> GenerateInput HKA:1349-1366 */
    int accelerate = 0, brake = 0; float turnInput = 0.0f;
    if (a->flags & HK_F_ACTIVE) {
        accelerate = (a->flags & HK_F_ACCEL) != 0; brake = (a->flags & HK_F_BRAKE) != 0; turnInput = a->steering;
    }
    (void)rl_agent; (void)act_steer; (void)act_branch;
    a->final_steer = kart_steer(e, a->acc_ang_v);                                       /* UpdateStats AK:295-302 */
    if (!(a->flags & HK_F_CAN_MOVE)) return;                                            /* AK:270 */
    /* MoveVehicle AK:363 */
    const float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
    float accelInput = (accelerate ? 1.0f : 0.0f) - (brake ? 1.0f : 0.0f);             /* :373 */
    float localVelZ = a->vx * fx + a->vz * fz;                                          /* :377 InverseTransformVector */
    int accelDirectionIsFwd = accelInput >= 0;
    int localVelDirectionIsFwd = localVelZ >= 0;
    float maxSpeed = localVelDirectionIsFwd ? st->TopSpeed : st->ReverseSpeed;          /* :383 */
    float wear = tire_wear_proportion(e, a->final_steer);
    float maxAllowedSpeed = sqrtf(max_lat_gs_for_wear(e, wear) * 9.81f * f_abs(turning_radius(a->vx, a->vz, fx, fz, a->wy)));  /* :384 */
    if (!(isinf(maxAllowedSpeed) || isnan(maxAllowedSpeed)))
        maxSpeed = f_clamp(maxSpeed, 0.001f, f_max(maxAllowedSpeed, 0.001f));           /* :388 */
    float accelPower = accelDirectionIsFwd ? st->Acceleration : st->ReverseAcceleration;
    float currentSpeed = mag3(a->vx, 0.0f, a->vz);                                      /* :392 */
    float accelRampT = currentSpeed / maxSpeed;
    float multipliedAccelerationCurve = st->AccelerationCurve * 5;                      /* :394 */
    float tt = accelRampT * accelRampT;
    tt = f_clamp(tt, 0.0f, 1.0f);                                                       /* Mathf.Lerp clamps t (Q10) */
    float accelRamp = multipliedAccelerationCurve + (1 - multipliedAccelerationCurve) * tt;
    int isBraking = (localVelDirectionIsFwd && brake) || (!localVelDirectionIsFwd && accelerate);   /* :397 */
    float finalAccelPower = isBraking ? st->Braking : accelPower;
    float finalAcceleration = finalAccelPower * accelRamp;
    float turningPower = turnInput * a->final_steer * (f_abs(currentSpeed) > 0.5f ? 1.0f : 0.0f);   /* :406 */
    float fwx = fx, fwz = fz;
    rot_y(turningPower * DEG2RAD_F, &fwx, &fwz);                                        /* :408-409 */
    float accx = fwx * accelInput * finalAcceleration * 1.0f;                           /* :410 (grounded) */
    float accz = fwz * accelInput * finalAcceleration * 1.0f;
    int wasOverMaxSpeed = currentSpeed >= maxSpeed;                                     /* :413 */
    if (wasOverMaxSpeed && !isBraking) { accx *= 0.0f; accz *= 0.0f; }
    float nvx = a->vx + accx * dt, nvz = a->vz + accz * dt;                             /* :419 */
    if (wasOverMaxSpeed) {                                                              /* :423-426 ClampMagnitude */
        float sq = nvx * nvx + nvz * nvz;
        if (sq > maxSpeed * maxSpeed) {
            float mg = sqrtf(sq);
            nvx = (nvx / mg) * maxSpeed; nvz = (nvz / mg) * maxSpeed;
        }
    }
    if (f_abs(accelInput) < 0.01f) {                                                    /* :429-432 coasting MoveTowards */
        float maxDelta = dt * st->CoastingDrag;
        float tx = 0.0f - nvx, tz = 0.0f - nvz;
        float sq = tx * tx + tz * tz;
        if (sq == 0.0f || sq <= maxDelta * maxDelta) { nvx = 0.0f; nvz = 0.0f; }
        else { float d = sqrtf(sq); nvx = nvx + tx / d * maxDelta; nvz = nvz + tz / d * maxDelta; }
    }
    a->vx = nvx; a->vz = nvz;                                                           /* :434 */
    float angularVelocitySteering = 0.4f;                                               /* :446 */
    if (!localVelDirectionIsFwd && !accelDirectionIsFwd) angularVelocitySteering *= -1.0f;   /* :450 */
    {   /* :456 Mathf.MoveTowards(angularVel.y, turningPower * steering, dt * 20) */
        float target = turningPower * angularVelocitySteering, maxDelta = dt * 20.0f;
        if (f_abs(target - a->wy) <= maxDelta) a->wy = target;
        else a->wy = a->wy + f_sign(target - a->wy) * maxDelta;
    }
    a->acc_ang_v += f_abs(a->wy);                                                       /* :457 */
    rot_y(turningPower * f_sign(localVelZ) * 25.0f * st->Grip * dt * DEG2RAD_F, &a->vx, &a->vz);   /* :466 */
}

/* ------------------------------------------------------------------ engine: KartAnimation steering + the WheelColliders' tire forces
 * (hk.h hk_engine_params).  The reference never calls this: Unity does, between two FixedUpdates.
 * Arithmetic contract (the kernels evaluate the same float expressions in the same order): divisions by configuration constants are
 * multiplications by reciprocals formed once (hko_engine_derive); a friction curve is ONE cubic in Horner form on its piece. */
static void curve_derive(hko_curve* c, float ext_slip, float ext_value, float asy_slip, float asy_value, float slope0)
{   /* WheelFrictionCurve: Hermite pieces through (0, 0) [slope slope0 * ext_value / ext_slip], (ext_slip, ext_value) and (asy_slip, asy_value),
     * flat at both knots and beyond.  Piece 1 in t = slip / ext_slip: E ((p - 2) t^3 + (3 - 2 p) t^2 + p t); piece 2 in
     * t = (slip - ext_slip) / (asy_slip - ext_slip): E + (A - E) (3 t^2 - 2 t^3) */
    c->ext = ext_slip; c->asy = asy_slip;
    c->inv_ext = 1.0f / ext_slip; c->inv_span = 1.0f / (asy_slip - ext_slip);
    c->a3 = ext_value * (slope0 - 2.0f); c->a2 = ext_value * (3.0f - 2.0f * slope0); c->a1 = ext_value * slope0;
    c->b3 = -2.0f * (asy_value - ext_value); c->b2 = 3.0f * (asy_value - ext_value); c->b0 = ext_value;
    c->flat = asy_value;
}
static inline float curve_eval(const hko_curve* c, float slip)
{
    const int in1 = slip <= c->ext;
    const float t = in1 ? slip * c->inv_ext : (slip - c->ext) * c->inv_span;
    const float c3 = in1 ? c->a3 : c->b3, c2 = in1 ? c->a2 : c->b2, c1 = in1 ? c->a1 : 0.0f, c0 = in1 ? 0.0f : c->b0;
    const float v = ((c3 * t + c2) * t + c1) * t + c0;
    return slip <= c->asy ? v : c->flat;
}
static void hko_engine_derive(const hk_config* cfg, hko_engine* E)
{
    const hk_engine_params* g = &cfg->engine;
    memset(E, 0, sizeof(*E));
    if (!(g->mass > 0.0f) || !(g->inertia_y > 0.0f)) return;
    E->inv_m = 1.0f / g->mass; E->inv_i = 1.0f / g->inertia_y;
    const float span = g->axle_zf - g->axle_zr;
    if (!g->wheel_friction || !(span > 0.0f)) return;
    const float load_f = (-g->axle_zr / span) * g->mass * g->gravity, load_r = (g->axle_zf / span) * g->mass * g->gravity;   /* static axle loads */
    E->side_kf = g->side_stiffness * load_f * cfg->dt; E->side_kr = g->side_stiffness * load_r * cfg->dt;
    E->jden_r = 1.0f / (E->inv_m + g->axle_zr * g->axle_zr * E->inv_i);
    curve_derive(&E->side, g->side_ext_slip, g->side_ext_value, g->side_asy_slip, g->side_asy_value, g->side_slope0);
    if (!g->wheel_rolling) return;
    E->inv_mw = 1.0f / g->wheel_mass;
    E->fwd_kf = g->fwd_stiffness * load_f * cfg->dt; E->fwd_kr = g->fwd_stiffness * load_r * cfg->dt;
    E->inv_damp_f = 1.0f / (1.0f + cfg->dt * g->wheel_damping / (0.5f * g->wheel_mass * g->wheel_radius_f * g->wheel_radius_f));
    E->inv_damp_r = 1.0f / (1.0f + cfg->dt * g->wheel_damping / (0.5f * g->wheel_mass * g->wheel_radius_r * g->wheel_radius_r));
    E->jlden_r = 1.0f / (E->inv_mw + E->inv_m);
    curve_derive(&E->fwd, g->fwd_ext_slip, g->fwd_ext_value, g->fwd_asy_slip, g->fwd_asy_value, g->side_slope0);
}

static void engine_wheels(const hko_env* e, hk_agent_state* a)
{
    const hk_engine_params* g = &e->cfg.engine;
    const float dt = e->cfg.dt;
    /* KartAnimation.cs:56: m_SmoothedSteeringInput = MoveTowards(., kartController.Input.TurnInput, steeringAnimationDamping * dt) */
    {
        const float turn = (a->flags & HK_F_ACTIVE) ? a->steering : 0.0f;
        const float maxDelta = g->steer_damping * dt;
        if (f_abs(turn - a->steer_smoothed) <= maxDelta) a->steer_smoothed = turn;
        else a->steer_smoothed = a->steer_smoothed + f_sign(turn - a->steer_smoothed) * maxDelta;
    }
    if (!g->wheel_friction || !(a->flags & HK_F_CAN_MOVE)) return;
    const hko_engine E = e->eng;                                       /* derived once in hko_create */
    const float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
    const float rx = fz, rz = -fx;                                     /* right */
    const float delta = a->steer_smoothed * g->max_steer_deg * DEG2RAD_F;   /* :59-63 steerAngle of both front wheels */
    const float sd = hk_sinf(delta), cd = hk_cosf(delta);
    float dvx = 0.0f, dvz = 0.0f, dw = 0.0f;
    /* ---- front axle: wheels turned by delta about y (positive = to the right); an impulse along the wheels' right axis at the axle has
     * the lever zf cos(delta) about y, one along their forward axis zf sin(delta) */
    {
        const float zk = g->axle_zf;
        const float wfx = cd * fx + sd * rx, wfz = cd * fz + sd * rz;
        const float wlx = cd * rx - sd * fx, wlz = cd * rz - sd * fz;
        const float vkx = a->vx + a->wy * zk * rx, vkz = a->vz + a->wy * zk * rz;        /* v + omega x r, r = zk * forward */
        const float vlong = vkx * wfx + vkz * wfz, vlat = vkx * wlx + vkz * wlz;
        const float slip = f_abs(vlat) / (f_abs(vlong) + g->slip_min_speed);
        float jn = curve_eval(&E.side, slip) * E.side_kf;
        const float lev = zk * cd;
        const float jmax = f_abs(vlat) / (E.inv_m + lev * lev * E.inv_i);                /* what stops the axle's sideways motion */
        if (jn > jmax) jn = jmax;
        if (vlat > 0.0f) jn = -jn;
        dvx += wlx * (jn * E.inv_m); dvz += wlz * (jn * E.inv_m); dw += jn * lev * E.inv_i;
        if (g->wheel_rolling) {
            /* the pair's rim speed u against the ground speed along the wheel: slip = (u - v_long) / (|v_long| + 4), the longitudinal force
             * forwardFriction(|slip|) * load on the body (forwards if the wheels turn faster), the opposite torque on the wheels, whose spin
             * wheelDampingRate also slows (implicitly, as PhysX integrates it).  A pair = 2 wheels: in u = omega r, du/dt = -F / m_wheel - ... */
            const float du = a->wheel_uf - vlong;
            const float ls = du / (f_abs(vlong) + g->long_slip_min_speed);
            float jl = curve_eval(&E.fwd, f_abs(ls)) * E.fwd_kf;
            const float levl = zk * sd;
            const float jlmax = f_abs(du) / (E.inv_mw + E.inv_m + levl * levl * E.inv_i);    /* what makes rim and ground speed equal */
            if (jl > jlmax) jl = jlmax;
            if (ls < 0.0f) jl = -jl;                                    /* wheels slower than the ground: the body is held back */
            dvx += wfx * (jl * E.inv_m); dvz += wfz * (jl * E.inv_m); dw += jl * levl * E.inv_i;
            a->wheel_uf = (a->wheel_uf - jl * E.inv_mw) * E.inv_damp_f;
        }
    }
    /* ---- rear axle (not steered): levers zr and 0 */
    {
        const float zk = g->axle_zr;
        const float vkx = a->vx + a->wy * zk * rx, vkz = a->vz + a->wy * zk * rz;
        const float vlong = vkx * fx + vkz * fz, vlat = vkx * rx + vkz * rz;
        const float slip = f_abs(vlat) / (f_abs(vlong) + g->slip_min_speed);
        float jn = curve_eval(&E.side, slip) * E.side_kr;
        const float jmax = f_abs(vlat) * E.jden_r;
        if (jn > jmax) jn = jmax;
        if (vlat > 0.0f) jn = -jn;
        dvx += rx * (jn * E.inv_m); dvz += rz * (jn * E.inv_m); dw += jn * zk * E.inv_i;
        if (g->wheel_rolling) {
            const float du = a->wheel_ur - vlong;
            const float ls = du / (f_abs(vlong) + g->long_slip_min_speed);
            float jl = curve_eval(&E.fwd, f_abs(ls)) * E.fwd_kr;
            const float jlmax = f_abs(du) * E.jlden_r;
            if (jl > jlmax) jl = jlmax;
            if (ls < 0.0f) jl = -jl;
            dvx += fx * (jl * E.inv_m); dvz += fz * (jl * E.inv_m);
            a->wheel_ur = (a->wheel_ur - jl * E.inv_mw) * E.inv_damp_r;
        }
    }
    a->vx += dvx; a->vz += dvz; a->wy += dw;
}

/* ------------------------------------------------------------------ trigger callbacks: HKA.OnTriggerEnter :611-675 */
static int calculate_lane(const hko_env* e, const hk_section* s, float px, float pz)
{   /* DPT.CalculateLane :116-148 (3-D distances, first minimum wins) */
    float dy = e->cfg.kart_y - s->marker_y;
    float d[4];
    for (int l = 0; l < 4; l++) d[l] = mag3(px - s->lane_x[l], dy, pz - s->lane_z[l]);
    float mn = f_min(f_min(d[0], d[1]), f_min(d[2], d[3]));
    for (int l = 0; l < 4; l++) if (mn == d[l]) return l + 1;
    return -1;
}

static void on_trigger_enter(hko_env* e, int env, int ai, int t)
{
    hk_env_state* es = &e->es[env];
    hk_agent_state* a = &e->ag[(size_t)env * e->A + ai];
    const int L = e->L, H = e->cfg.section_horizon;
    if (!(a->flags & HK_F_ACTIVE)) return;                                              /* :613 */
    /* KA.FindSectionIndex :348-364 */
    int index = -1, lane = -1;
    int lo = a->section_index - H; if (lo < a->init_checkpoint_index) lo = a->init_checkpoint_index;
    for (int i = lo; i < a->section_index + H; i++) {
        int idx = i < 0 ? i + L : i;
        if (idx % L == t) { index = idx; lane = calculate_lane(e, &e->sec[idx % L], a->px, a->pz); break; }
    }
    const int sec = a->section_index;
    const int rw = e->cfg.rewards;
    float lane_div = 1.0f, vel_div = 1.0f;                                              /* :618-619 */
    if (index != -1 && ((index > sec) || (index % L == 0 && sec % L == L - 1))) {       /* :621 */
        const int key = index % L;
        if (a->plan_lane[key] != 0) {                                                   /* :623 */
            if (rw) hko_rw_dividers(e, ai, a, index, lane, &lane_div, &vel_div);        /* :625-626 */
            /* UpdateLaneDifferenceCalculation KA:226-230 */
            pt2 lm = lane_marker(e, key, a->plan_lane[key]);
            float dist = mag3(a->px - lm.x, e->cfg.kart_y - e->sec[key].marker_y, a->pz - lm.z);
            a->avg_lane_diff = (f_max(dist - 1.3f, 0.0f) + a->avg_lane_diff * (index - a->init_checkpoint_index - 1)) /
                               (index - a->init_checkpoint_index);
            /* UpdateVelocityDifferenceCalculation KA:235-239 */
            float velocity = mag3(a->vx, 0.0f, a->vz);
            a->avg_vel_diff = ((velocity - a->plan_vel[key]) + a->avg_vel_diff * (index - a->init_checkpoint_index - 1)) /
                              (index - a->init_checkpoint_index);
            a->plan_lane[key] = 0; a->plan_vel[key] = 0.0f;                             /* :629-630 */
        }
        int dl = a->lane - lane; if (dl < 0) dl = -dl;
        if (a->lane_changes + dl > e->cfg.max_lane_changes && is_straight(e, sec)) {    /* :636-640 */
            if (rw) hko_rw_swerve(e, a);
            a->illegal_lane_changes += 1;
        }
        if (is_straight(e, sec) != is_straight(e, index)) a->lane_changes = 0;          /* :641 */
        else if (a->lane != lane) a->lane_changes += dl;                                /* :645 */
        a->section_index = index; a->lane = lane;                                       /* :649-650 */
        if (e->mcts) {
            hk_mcts_state* m = &e->mcts[(size_t)env * e->A + ai];
            m->sec_time[index & (HK_MCTS_SECTIME_RING - 1)] = es->episode_steps;            /* :651 sectionTimes */
            m->root_live = 0; m->root_cycles = 0;                                           /* :660-661 currentRoot = null */
        }
        const int goal = e->cfg.laps * L + 1;                                           /* REC:165 */
        if (a->section_index == goal) {                                                 /* :652 -> REC.ResolveEvent :469-474 */
            a->time_steps = es->episode_steps;
            if (rw) hko_rw_section(e, env, ai, lane_div, vel_div);                      /* REC:471 */
            deactivate(e, a);
            es->inactive_mask |= 1u << ai;
        } else if (rw) hko_rw_section(e, env, ai, lane_div, vel_div);                   /* REC:467 */
    } else if (index != -1 && ((index < sec) || (sec % L == 0 && index % L == L - 1))) {/* :663 */
        if (rw) hko_rw_reverse(e, a, sec, index);                                       /* :666 */
        a->section_index = index;                                                       /* :667 */
        if (e->mcts) { hk_mcts_state* m = &e->mcts[(size_t)env * e->A + ai]; m->root_live = 0; m->root_cycles = 0; }   /* :668-669 */
    } else if (index == -1) {                                                           /* :671 DroveReverseLimit REC:475-479 */
        a->time_steps = e->cfg.max_episode_steps * 6;
        deactivate(e, a);
        es->inactive_mask |= 1u << ai;
    }
}

/* does the kart's capsule overlap section t's Trigger box?  (capsule AABB in the box frame vs the box; exact while
 * the capsule's lateral range lies inside the 10 m wide box, which the 9.2 m wide walled track guarantees) */
static int overlaps_trigger(const hko_env* e, const hk_agent_state* a, int t)
{
    const hk_section* s = &e->sec[t];
    const sec_pre* sp = &e->sp[t];
    float ax, az, bx, bz;
    kart_core(a, a->px, a->pz, &ax, &az, &bx, &bz);
    float rax = ax - s->trig_x, raz = az - s->trig_z, rbx = bx - s->trig_x, rbz = bz - s->trig_z;
    float lax = rax * sp->fz - raz * sp->fx, laz = rax * sp->fx + raz * sp->fz;
    float lbx = rbx * sp->fz - rbz * sp->fx, lbz = rbx * sp->fx + rbz * sp->fz;
    float zlo = f_min(laz, lbz) - KART_CAP_R, zhi = f_max(laz, lbz) + KART_CAP_R;
    float xlo = f_min(lax, lbx) - KART_CAP_R, xhi = f_max(lax, lbx) + KART_CAP_R;
    return zlo <= TRIG_HZ && zhi >= -TRIG_HZ && xlo <= TRIG_HX && xhi >= -TRIG_HX;
}

static void telemetry_update(const hko_env* e, const hk_env_state* es, hk_agent_state* a);

/* ------------------------------------------------------------------ one tick of one env */
static void finish_episode(hko_env* e, int env, int timeout)
{
    hk_env_state* es = &e->es[env];
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    for (int i = 0; i < e->A; i++)
        if (ags[i].flags & HK_F_ACTIVE) deactivate(e, &ags[i]);                         /* REC:243-247 / :284-288 */
    if (es->initial_started || timeout) {
        snapshot_results(e, env);                                                       /* telemetry block REC:249-265 */
        es->episodes_done += 1;
        es->status = (es->status & ~2u) | (timeout ? 2u : 0u);
        es->experiment_num += 1;                                                        /* REC:268-269 / :308 */
    }
    if (e->cfg.rewards) {                                                               /* REC:267 / :306, before ResetGame */
        hko_rw_goal_timing(e, env);
        for (int i = 0; i < e->A; i++) e->res[(size_t)env * e->A + i].group_reward = ags[i].group_reward;
    }
    reset_env(e, env);                                                                  /* REC:270 / :309 */
    es->initial_started = 1;                                                            /* REC:277 */
}

static void step_env(hko_env* e, int env)
{
    const hk_config* cfg = &e->cfg;
    hk_env_state* es = &e->es[env];
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    const int A = e->A, L = e->L;
    const uint32_t all_mask = (1u << A) - 1u;
    /* Academy step: KartAgent.OnActionReceived rewards on the state the previous tick left (a parked env has no agents) */
    if (cfg->rewards && !(!cfg->auto_reset && (es->inactive_mask & all_mask) == all_mask)) hko_rw_academy(e, env);
    /* (a) REC.FixedUpdate :239-311 */
    if ((es->inactive_mask & all_mask) == all_mask) {
        if (!cfg->auto_reset) {
            if (!(es->status & 4u)) { snapshot_results(e, env); es->episodes_done += 1; es->status |= 4u; }
            return;
        }
        finish_episode(e, env, 0);
    } else {
        es->episode_steps += 1;                                                         /* :281 */
        if (es->episode_steps >= cfg->max_episode_steps) {                              /* :282 */
            if (!cfg->auto_reset) {
                for (int i = 0; i < A; i++) if (ags[i].flags & HK_F_ACTIVE) deactivate(e, &ags[i]);
                es->inactive_mask = all_mask;
                snapshot_results(e, env);
                es->episodes_done += 1; es->status |= 2u | 4u;
                return;
            }
            finish_episode(e, env, 1);
        }
    }
    /* StartRaceAfterDelay REC:721-744: WaitForSeconds(1.5) = start_hold_ticks ticks after the reset */
    if (es->episode_steps >= cfg->start_hold_ticks)
        for (int i = 0; i < A; i++)
            if ((ags[i].flags & HK_F_ACTIVE) && !(ags[i].flags & HK_F_CAN_MOVE)) ags[i].flags |= HK_F_CAN_MOVE;
    /* (b) agents */
    hk_agent_state snap[HK_MAX_AGENTS];
    (void)snap;
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED)) continue;            /* disabled GameObject: no FixedUpdate */
        /* KA.FixedUpdate :135-167 */
        static const int csens[3] = {0, 1, 5};
        static const float clen[3] = {0.8f, 0.9f, 0.9f};
        int hitAgent = 0;
        for (int q = 0; q < 3; q++) {
            float ox, oz, rdx, rdz;
            sensor_ray(e, a, csens[q], &ox, &oz, &rdx, &rdz);
            hitAgent |= raycast_agents(e, ags, i, ox, oz, rdx, rdz, clen[q], NULL) >= 0.0f;
        }
        const int fc = (a->flags & HK_F_FORWARD_COLLISION) != 0;
        if (hitAgent && !fc && (a->last_collision_time == 0 || es->episode_steps - a->last_collision_time > 75)) {   /* :150 */
            a->flags |= HK_F_FORWARD_COLLISION; a->forward_collisions += 1; a->last_collision_time = es->episode_steps;
        } else if (hitAgent) {
            a->flags |= HK_F_FORWARD_COLLISION; a->last_collision_time = es->episode_steps;
        } else {
            a->flags &= ~HK_F_FORWARD_COLLISION;
        }
        if (cfg->rewards) hko_rw_not_at_goal(e, a);                                     /* :165 */
    }
    /* HKA.FixedUpdate :313-362. SolveLQR only writes the ego's own controls, and reads poses/plans that no agent
     * script modifies before (c), so the A solves of a tick are order independent (Q9). */
    const int inactive_before = es->inactive_mask;
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED)) continue;
        const int inactive = (inactive_before >> i) & 1;
        if ((es->episode_steps % (A > 2 ? 4 : 1)) == 0 && cfg->low_mode[i] == HK_LOW_LQR && !inactive)   /* :317-323 */
            solve_lqr(e, env, i);
        if (cfg->low_mode[i] == HK_LOW_RL && (a->flags & HK_F_ACTIVE)) {                /* KA.OnActionReceived :440, HKA:1371-1379 */
            a->steering = e->act_steer[(size_t)env * A + i];
            int br = e->act_branch[(size_t)env * A + i];
            if (br > 1) a->flags |= HK_F_ACCEL; else a->flags &= ~HK_F_ACCEL;
            if (br < 1) a->flags |= HK_F_BRAKE; else a->flags &= ~HK_F_BRAKE;
        }
        if (es->episode_steps % 100 == 0 && es->episode_steps < cfg->max_episode_steps && es->episode_steps > 0 && !inactive) {
            if (cfg->training_agent[i]) plan_randomly(e, env, i, a);                    /* :357-360 Mode == Training */
            else if (cfg->high_mode[i] == HK_HIGH_FIXED) plan_fixed(e, i, a);           /* :331-355 */
            else hko_mcts_request(e, env, i, cfg->mcts_iterations, cfg->mcts_latency_ticks, NULL);   /* :335-350 planWithMCTS() */
        }
        if (cfg->high_mode[i] == HK_HIGH_MCTS) {                                        /* :366-402 */
            hk_mcts_state* m = &e->mcts[(size_t)env * A + i];
            if (m->ready_step >= 0 && es->episode_steps >= m->ready_step) hko_mcts_promote(m);
            hko_mcts_consume(e, env, i);
        }
    }
    /* (c) ArcadeKart.FixedUpdate */
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED)) continue;
        arcade_kart_update(e, a, 0, 0.0f, 1);
    }
    /* (d) engine restatement --------------------------------------------------------------- */
    const float dt = cfg->dt;
    /* KartAnimation.FixedUpdate (execution order 100: after ArcadeKart; KartAnimation.cs:54-63) steers the front WheelColliders,
     * then the engine applies the four wheels' sideways friction as forces on the rigid body */
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED)) continue;
        engine_wheels(e, a);
    }
    /* integrate (semi-implicit Euler; angular damping 1 - angularDrag*dt as PhysX applies it) */
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED) || !(a->flags & HK_F_CAN_MOVE)) continue;       /* constraints FreezeAll while held */
        a->wy = a->wy * (1.0f - cfg->stats.AngularDrag * dt);
        a->yaw = a->yaw + a->wy * dt;
        if (a->yaw < 0.0f) a->yaw += TWO_PI_F;
        if (a->yaw >= TWO_PI_F) a->yaw -= TWO_PI_F;
        a->px = a->px + a->vx * dt;
        a->pz = a->pz + a->vz * dt;
    }
    /* kart-kart contacts: every pair from ONE snapshot (Jacobi), corrections summed in partner order */
    {
        float cpx[HK_MAX_AGENTS], cpz[HK_MAX_AGENTS], cvx[HK_MAX_AGENTS], cvz[HK_MAX_AGENTS], cwy[HK_MAX_AGENTS];
        int touched[HK_MAX_AGENTS];
        for (int i = 0; i < A; i++) { cpx[i] = 0; cpz[i] = 0; cvx[i] = 0; cvz[i] = 0; cwy[i] = 0; touched[i] = 0; }
        for (int i = 0; i < A; i++) {
            if (!(ags[i].flags & HK_F_ENABLED)) continue;
            float ax, az, bx, bz;
            kart_core(&ags[i], ags[i].px, ags[i].pz, &ax, &az, &bx, &bz);
            for (int j = 0; j < A; j++) {
                if (j == i || !(ags[j].flags & HK_F_ENABLED)) continue;
                float cx, cz, dx, dz, c1x, c1z, c2x, c2z;
                kart_core(&ags[j], ags[j].px, ags[j].pz, &cx, &cz, &dx, &dz);
                float d2 = seg_seg_closest(ax, az, bx, bz, cx, cz, dx, dz, &c1x, &c1z, &c2x, &c2z);
                const float rr = 2.0f * KART_CAP_R;
                if (d2 < rr * rr) {
                    float d = sqrtf(d2);
                    float nx, nz;
                    if (d > 1e-6f) { nx = (c1x - c2x) / d; nz = (c1z - c2z) / d; }
                    else {
                        float ex = ags[i].px - ags[j].px, ez = ags[i].pz - ags[j].pz;
                        float el = sqrtf(ex * ex + ez * ez);
                        if (el > 1e-6f) { nx = ex / el; nz = ez / el; } else { nx = (i < j) ? 1.0f : -1.0f; nz = 0.0f; }
                    }
                    float pen = rr - d;
                    /* a held (frozen) kart does not move: the free one takes the whole correction */
                    float share = (ags[j].flags & HK_F_CAN_MOVE) ? 0.5f : 1.0f;
                    cpx[i] += nx * (pen * share); cpz[i] += nz * (pen * share);
                    if (!cfg->engine.contact_yaw) {
                        float vrel = (ags[i].vx - ags[j].vx) * nx + (ags[i].vz - ags[j].vz) * nz;
                        if (vrel < 0.0f) { cvx[i] -= nx * (vrel * share); cvz[i] -= nz * (vrel * share); }
                    } else {
                        /* frictionless inelastic contact of two free rigid bodies at the surface points (c1 - R n on i, c2 + R n on j):
                         * the impulse along n that stops the approach of the two points, acting off-centre on both */
                        const float inv_m = 1.0f / cfg->engine.mass, inv_i = 1.0f / cfg->engine.inertia_y;
                        const float rix = (c1x - nx * KART_CAP_R) - ags[i].px, riz = (c1z - nz * KART_CAP_R) - ags[i].pz;
                        const float rjx = (c2x + nx * KART_CAP_R) - ags[j].px, rjz = (c2z + nz * KART_CAP_R) - ags[j].pz;
                        const float ki = riz * nx - rix * nz, kj = rjz * nx - rjx * nz;
                        float vrel = (ags[i].vx - ags[j].vx) * nx + (ags[i].vz - ags[j].vz) * nz + ags[i].wy * ki - ags[j].wy * kj;
                        if (vrel < 0.0f) {
                            float den = inv_m + ki * ki * inv_i;
                            if (ags[j].flags & HK_F_CAN_MOVE) den = den + (inv_m + kj * kj * inv_i);
                            const float jn = -vrel / den;
                            cvx[i] += nx * (jn * inv_m); cvz[i] += nz * (jn * inv_m); cwy[i] += jn * ki * inv_i;
                        }
                    }
                    touched[i] = 1;
                }
            }
        }
        for (int i = 0; i < A; i++) {
            hk_agent_state* a = &ags[i];
            if (!touched[i] || !(a->flags & HK_F_CAN_MOVE)) continue;
            a->px += cpx[i]; a->pz += cpz[i]; a->vx += cvx[i]; a->vz += cvz[i]; a->wy += cwy[i];
        }
        for (int i = 0; i < A; i++) if (touched[i]) ags[i].flags |= HK_F_HAS_COLLISION; else ags[i].flags &= ~HK_F_HAS_COLLISION;
    }
    /* kart-wall contacts: resolve the deepest penetration, twice (corners) */
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED) || !(a->flags & HK_F_CAN_MOVE)) continue;
        for (int pass = 0; pass < 2; pass++) {
            float ax, az, bx, bz;
            kart_core(a, a->px, a->pz, &ax, &az, &bx, &bz);
            float bestpen = 0.0f, bnx = 0.0f, bnz = 0.0f, bcx = 0.0f, bcz = 0.0f;
            int found = 0;
            for (int w = 0; w < e->NW; w++) {
                const hk_wall_seg* ws = &e->walls[w];
                float c1x, c1z, c2x, c2z;
                float d2 = seg_seg_closest(ax, az, bx, bz, ws->x0, ws->z0, ws->x1, ws->z1, &c1x, &c1z, &c2x, &c2z);
                if (d2 < KART_CAP_R * KART_CAP_R) {
                    float d = sqrtf(d2);
                    float pen = KART_CAP_R - d;
                    float nx, nz;
                    if (d > 1e-6f) { nx = (c1x - c2x) / d; nz = (c1z - c2z) / d; }
                    else {
                        /* core touches the wall: use the wall normal pointing at the kart centre */
                        float ex = ws->x1 - ws->x0, ez = ws->z1 - ws->z0;
                        float el = sqrtf(ex * ex + ez * ez);
                        nx = -ez / el; nz = ex / el;
                        if ((a->px - ws->x0) * nx + (a->pz - ws->z0) * nz < 0.0f) { nx = -nx; nz = -nz; }
                    }
                    if (!found || pen > bestpen) { found = 1; bestpen = pen; bnx = nx; bnz = nz; bcx = c1x; bcz = c1z; }
                }
            }
            if (!found) break;
            if (!cfg->engine.contact_yaw) {
                float vn = a->vx * bnx + a->vz * bnz;
                if (vn < 0.0f) { a->vx -= bnx * vn; a->vz -= bnz * vn; }
            } else {
                /* frictionless inelastic contact at the capsule's surface point c1 - R n (lever arm about the centre of mass = the
                 * kart origin, taken before the push-out): the impulse stops the POINT's approach and turns the kart along the wall */
                const float inv_m = 1.0f / cfg->engine.mass, inv_i = 1.0f / cfg->engine.inertia_y;
                const float rx = (bcx - bnx * KART_CAP_R) - a->px, rz = (bcz - bnz * KART_CAP_R) - a->pz;
                const float k = rz * bnx - rx * bnz;
                const float vn = a->vx * bnx + a->vz * bnz + a->wy * k;
                if (vn < 0.0f) {
                    const float jn = -vn / (inv_m + k * k * inv_i);
                    a->vx += bnx * (jn * inv_m); a->vz += bnz * (jn * inv_m); a->wy += jn * k * inv_i;
                }
            }
            a->px += bnx * bestpen; a->pz += bnz * bestpen;
            a->flags |= HK_F_HAS_COLLISION;
            a->contact_nx = bnx; a->contact_nz = bnz;
        }
    }
    /* NaN / Inf guard word */
    for (int i = 0; i < A; i++) {
        const hk_agent_state* a = &ags[i];
        if (!isfinite(a->px) || !isfinite(a->pz) || !isfinite(a->vx) || !isfinite(a->vz) || !isfinite(a->yaw) || !isfinite(a->wy))
            es->status |= 1u;
    }
    /* trigger dispatch: OnTriggerEnter for every Trigger box newly overlapped, ascending section order */
    for (int i = 0; i < A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED)) continue;
        uint32_t lo = 0, hi = 0;
        for (int t = 0; t < L; t++)
            if (overlaps_trigger(e, a, t)) { if (t < 32) lo |= 1u << t; else hi |= 1u << (t - 32); }
        uint32_t nlo = lo & ~a->trig_lo, nhi = hi & ~a->trig_hi;
        a->trig_lo = lo; a->trig_hi = hi;
        for (int t = 0; t < L; t++) {
            int ent = t < 32 ? (nlo >> t) & 1 : (nhi >> (t - 32)) & 1;
            if (ent) on_trigger_enter(e, env, i, t);
        }
    }
    /* TelemetryViewer.Update (once per tick here; the reference runs it once per rendered frame) */
    for (int i = 0; i < A; i++) telemetry_update(e, es, &ags[i]);
}

/* TelemetryViewer.Update :49-88 for one agent (the viewer keeps its arrays across episodes; a lower lap count = reset) */
static void telemetry_update(const hko_env* e, const hk_env_state* es, hk_agent_state* a)
{
    const float dt = e->cfg.dt;
    int currentLap = a->section_index / e->L;                                           /* :58 */
    if (currentLap > a->tele_completed_laps) {                                          /* :59-68 */
        a->tele_completed_laps = currentLap;
        a->tele_last_lap = dt * (es->episode_steps - a->tele_lap_end_step);
        if (a->tele_best_lap < 10 || a->tele_last_lap < a->tele_best_lap) a->tele_best_lap = a->tele_last_lap;
        a->tele_lap_end_step = es->episode_steps;
    } else if (currentLap < a->tele_completed_laps) {                                   /* :69-75 */
        a->tele_completed_laps = currentLap;
        a->tele_last_lap = 0.0f; a->tele_best_lap = 0.0f; a->tele_lap_end_step = 0;
    }
    if (a->flags & HK_F_ACTIVE) a->tele_total_time = es->episode_steps * dt;            /* :76-79 */
}

/* ------------------------------------------------------------------ API */
static int next_perm(int* a, int n)
{
    int i = n - 2;
    while (i >= 0 && a[i] > a[i + 1]) i--;
    if (i < 0) return 0;
    int j = n - 1;
    while (a[j] < a[i]) j--;
    int t = a[i]; a[i] = a[j]; a[j] = t;
    for (int l = i + 1, r = n - 1; l < r; l++, r--) { t = a[l]; a[l] = a[r]; a[r] = t; }
    return 1;
}

hko_env* hko_create(const hk_config* cfg)
{
    if (!cfg || cfg->num_agents < 1 || cfg->num_agents > HK_MAX_AGENTS || cfg->num_sections < 1 || cfg->num_sections > HK_MAX_SECTIONS)
        return NULL;
    hko_env* e = (hko_env*)calloc(1, sizeof(*e));
    e->cfg = *cfg;
    hko_engine_derive(&e->cfg, &e->eng);
    e->E = cfg->num_envs; e->A = cfg->num_agents; e->L = cfg->num_sections; e->NW = cfg->num_walls;
    e->sec = (hk_section*)malloc(sizeof(hk_section) * e->L);
    memcpy(e->sec, cfg->sections, sizeof(hk_section) * e->L);
    e->walls = (hk_wall_seg*)malloc(sizeof(hk_wall_seg) * (e->NW > 0 ? e->NW : 1));
    memcpy(e->walls, cfg->walls, sizeof(hk_wall_seg) * e->NW);
    e->cfg.sections = e->sec; e->cfg.walls = e->walls;
    e->sp = (sec_pre*)malloc(sizeof(sec_pre) * e->L);
    for (int i = 0; i < e->L; i++) {
        e->sp[i].yaw_rad = e->sec[i].yaw_deg * DEG2RAD_F;
        e->sp[i].fx = hk_sinf(e->sp[i].yaw_rad);
        e->sp[i].fz = hk_cosf(e->sp[i].yaw_rad);
    }
    size_t na = (size_t)e->E * e->A;
    e->ag = (hk_agent_state*)calloc(na, sizeof(hk_agent_state));
    e->es = (hk_env_state*)calloc(e->E, sizeof(hk_env_state));
    e->res = (hk_episode_result*)calloc(na, sizeof(hk_episode_result));
    e->dbg = (hk_lq_debug*)calloc(na, sizeof(hk_lq_debug));
    e->act_steer = (float*)calloc(na, sizeof(float));
    e->act_branch = (int32_t*)calloc(na, sizeof(int32_t));
    if (cfg->rewards) {
        const size_t n = na * (size_t)hko_rw_table_len(e);
        e->sec_min_time = (int32_t*)malloc(n * sizeof(int32_t));
        e->sec_count = (uint8_t*)malloc(n);
        memset(e->sec_min_time, 0xFF, n * sizeof(int32_t)); memset(e->sec_count, 0, n);
    }
    for (int i = 0; i < e->A; i++)
        if (cfg->high_mode[i] == HK_HIGH_MCTS && !e->mcts) {
            e->mcts = (hk_mcts_state*)calloc(na, sizeof(hk_mcts_state));
            e->trees = (struct hko_tree*)calloc(na, hko_mcts_tree_bytes());
        }
    for (size_t i = 0; i < na; i++) { e->res[i].episode = -1; e->act_branch[i] = 1; }
    const hk_kart_stats* s = &cfg->stats;
    e->max_speed = f_max(s->TopSpeed, s->ReverseSpeed);                                  /* AK:210 */
    /* REC:588 with genTWP = 0.25 (Experiment / Race, REC:501-502) */
    e->init_acc_ang_v = -s->TireWearRate * hk_logf(1 - ((s->MaxSteer - s->MinSteer) * 0.25f / s->MaxSteer));
    /* kart capsule sliced at the sensor-ray height */
    {
        float dyc = SENSOR_LY - CAP_CENTER_LY;
        e->ray_agent_r = sqrtf(KART_CAP_R * KART_CAP_R - dyc * dyc);
    }
    for (int i = 0; i < HK_NUM_SENSORS; i++) {
        float d = cfg->sensor_yaw_deg[i] * DEG2RAD_F;
        e->sens_c[i] = hk_cosf(d); e->sens_s[i] = hk_sinf(d);
    }
    e->nperm = 1;
    for (int i = 2; i <= e->A; i++) e->nperm *= i;
    e->perms = (int*)malloc(sizeof(int) * e->nperm * e->A);
    int cur[HK_MAX_AGENTS];
    for (int i = 0; i < e->A; i++) cur[i] = i;
    int p = 0;
    do { memcpy(&e->perms[(size_t)p * e->A], cur, sizeof(int) * e->A); p++; } while (next_perm(cur, e->A));
    /* REC.Start :148-168: every agent starts inactive, so the first FixedUpdate resets (REC:241) */
    for (int i = 0; i < e->E; i++) e->es[i].inactive_mask = (1u << e->A) - 1u;
    return e;
}

void hko_destroy(hko_env* e)
{
    if (!e) return;
    free(e->sec); free(e->sp); free(e->walls); free(e->ag); free(e->es); free(e->res); free(e->dbg);
    hko_policy_free(e);
    hko_mcts_trees_free(e);
    free(e->mcts);
    free(e->sec_min_time); free(e->sec_count);
    free(e->act_steer); free(e->act_branch); free(e->perms); free(e);
}

int hko_reset(hko_env* e, const int32_t* env_ids, int n, int experiment_num)
{
    if (!e) return HK_ERR_INVALID;
    int cnt = env_ids ? n : e->E;
    for (int q = 0; q < cnt; q++) {
        int env = env_ids ? env_ids[q] : q;
        if (env < 0 || env >= e->E) return HK_ERR_INVALID;
        e->es[env].experiment_num = experiment_num >= 0 ? experiment_num : (e->cfg.env_id_base + env) % e->nperm;
        e->es[env].status = 0;
        e->es[env].initial_started = 1;
        reset_env(e, env);
        hko_policy_invalidate(e, env);
    }
    return 0;
}

/* host threads hko_step spreads the envs over (bench.py's cpu_baseline times 1 thread and all cores); returns the count in use */
int hko_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int hko_step(hko_env* e, int n_ticks)
{
    if (!e || n_ticks < 0) return HK_ERR_INVALID;
    if (e->n_policies == 0) {
#pragma omp parallel for schedule(dynamic, 8)
        for (int env = 0; env < e->E; env++)
            for (int t = 0; t < n_ticks; t++) step_env(e, env);
        e->academy_step += n_ticks;
        return 0;
    }
    /* with policies attached the Academy steps first in every tick (decision -> OnActionReceived), then the scripts */
    for (int t = 0; t < n_ticks; t++) {
        hko_policy_decide(e);
#pragma omp parallel for schedule(dynamic, 8)
        for (int env = 0; env < e->E; env++) step_env(e, env);
        e->academy_step += 1;
    }
    return 0;
}

int hko_set_actions(hko_env* e, const float* steer, const int32_t* branch)
{
    size_t na = (size_t)e->E * e->A;
    memcpy(e->act_steer, steer, na * sizeof(float));
    memcpy(e->act_branch, branch, na * sizeof(int32_t));
    return 0;
}
int hko_get_agent_state(hko_env* e, hk_agent_state* out) { memcpy(out, e->ag, sizeof(hk_agent_state) * e->E * e->A); return 0; }
int hko_set_agent_state(hko_env* e, const hk_agent_state* in) { memcpy(e->ag, in, sizeof(hk_agent_state) * e->E * e->A); return 0; }
int hko_get_env_state(hko_env* e, hk_env_state* out) { memcpy(out, e->es, sizeof(hk_env_state) * e->E); return 0; }
int hko_set_env_state(hko_env* e, const hk_env_state* in) { memcpy(e->es, in, sizeof(hk_env_state) * e->E); return 0; }
int hko_get_mcts_state(hko_env* e, hk_mcts_state* out)
{
    if (e->mcts) memcpy(out, e->mcts, sizeof(hk_mcts_state) * e->E * e->A);
    else memset(out, 0, sizeof(hk_mcts_state) * e->E * e->A);
    return 0;
}
int hko_get_episode_results(hko_env* e, hk_episode_result* out) { memcpy(out, e->res, sizeof(hk_episode_result) * e->E * e->A); return 0; }
int hko_debug_last_game(hko_env* e, int env, int ego, hk_lq_debug* out)
{
    if (!e || env < 0 || env >= e->E || ego < 0 || ego >= e->A) return HK_ERR_INVALID;
    *out = e->dbg[(size_t)env * e->A + ego];
    return 0;
}

/* ------------------------------------------------------------------ HKA.CollectObservations :485-604 */
static inline void inv_transform_point(const hk_agent_state* a, float wx, float wy, float wz, float ky, float out[3])
{   /* Transform.InverseTransformPoint for a yaw-only transform (Q10) */
    float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
    float rx = wx - a->px, rz = wz - a->pz;
    out[0] = rx * fz + rz * (-fx);
    out[1] = wy - ky;
    out[2] = rx * fx + rz * fz;
}
static float local_speed(const hko_env* e, const hk_agent_state* a)
{   /* AK:325-342 */
    if (!(a->flags & HK_F_CAN_MOVE)) return 0.0f;
    float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
    float dot = fx * a->vx + fz * a->vz;
    if (f_abs(dot) > 0.1f) {
        float speed = mag3(a->vx, 0.0f, a->vz);
        return dot < 0 ? -(speed / e->cfg.stats.ReverseSpeed) : (speed / e->cfg.stats.TopSpeed);
    }
    return 0.0f;
}

int hko_obs_dim(const hko_env* e) { return HK_NUM_SENSORS + e->cfg.section_horizon * 5 + 8 + 12 * (e->A - 1); }   /* HKA:424 */

void hko_observe_env(hko_env* e, int env, float* obs)
{
    const hk_config* cfg = &e->cfg;
    const int A = e->A, L = e->L, H = cfg->section_horizon;
    const int dim = hko_obs_dim(e);
    const int goal = cfg->laps * L + 1;
    {
        hk_agent_state* ags = &e->ag[(size_t)env * A];
        for (int i = 0; i < A; i++) {
            const hk_agent_state* a = &ags[i];
            float* o = obs + (size_t)i * dim;
            int p = 0;
            o[p++] = local_speed(e, a);                                                 /* :489 */
            o[p++] = (a->flags & HK_F_ACCEL) ? 1.0f : 0.0f;                             /* :490 bool -> 1/0 */
            o[p++] = (float)a->lane;                                                    /* :491 */
            o[p++] = a->lane_changes * 1.0f / cfg->max_lane_changes;                    /* :492 */
            o[p++] = (a->flags & HK_F_ACTIVE) ? 1.0f : 0.0f;                            /* :493 */
            o[p++] = a->section_index * 1.0f / goal;                                    /* :494 */
            o[p++] = is_straight(e, a->section_index) ? 1.0f : 0.0f;                    /* :495 */
            o[p++] = tire_wear_proportion(e, a->final_steer);                           /* :496 */
            for (int pass = 0; pass < 2; pass++) {                                      /* :500-527 team, then others */
                int cnt = pass == 0 ? cfg->n_team[i] : cfg->n_other[i];
                for (int j = 0; j < cnt; j++) {
                    const hk_agent_state* b = &ags[pass == 0 ? cfg->team_agents[i][j] : cfg->other_agents[i][j]];
                    o[p++] = local_speed(e, b);
                    o[p++] = (b->flags & HK_F_ACCEL) ? 1.0f : 0.0f;
                    o[p++] = (float)b->lane;
                    o[p++] = b->lane_changes * 1.0f / cfg->max_lane_changes;
                    o[p++] = (b->flags & HK_F_ACTIVE) ? 1.0f : 0.0f;
                    o[p++] = is_straight(e, b->section_index) ? 1.0f : 0.0f;
                    o[p++] = tire_wear_proportion(e, b->final_steer);
                    o[p++] = b->section_index * 1.0f / goal;
                    o[p++] = mag3(b->px - a->px, 0.0f, b->pz - a->pz);
                    float lp[3];
                    inv_transform_point(a, b->px, cfg->kart_y, b->pz, cfg->kart_y, lp);
                    o[p++] = lp[0]; o[p++] = lp[1]; o[p++] = lp[2];
                }
            }
            for (int s = a->section_index + 1; s < a->section_index + 1 + H; s++) {     /* :530-552 */
                int next = s % L;
                float lp[3];
                if (a->plan_lane[next] != 0) {
                    pt2 m = lane_marker(e, next, a->plan_lane[next]);
                    inv_transform_point(a, m.x, e->sec[next].marker_y, m.z, cfg->kart_y, lp);
                    o[p++] = lp[0]; o[p++] = lp[1]; o[p++] = lp[2];
                    o[p++] = a->plan_vel[next] / e->max_speed;
                } else {
                    inv_transform_point(a, e->sec[next].trig_x, e->sec[next].marker_y, e->sec[next].trig_z, cfg->kart_y, lp);
                    o[p++] = lp[0]; o[p++] = lp[1]; o[p++] = lp[2];
                    o[p++] = 1.0f;
                }
                o[p++] = is_straight(e, next) ? 1.0f : 0.0f;
            }
            for (int si = 0; si < HK_NUM_SENSORS; si++) {                               /* :553-603 */
                float ox, oz, dx, dz;
                sensor_ray(e, a, si, &ox, &oz, &dx, &dz);
                float ht = hko_raycast_track(e, ox, oz, dx, dz, cfg->ray_distance[si]);
                int who = -1;
                float ha = (a->flags & HK_F_ENABLED) ? raycast_agents(e, ags, i, ox, oz, dx, dz, cfg->ray_distance[si], &who) : -1.0f;
                if (ht >= 0.0f && (ha < 0.0f || ht < ha)) {                             /* :580-588 */
                    if (cfg->rewards && ht < cfg->wall_hit_validation[si]) hko_rw_hit(e, env, i, -1);
                    o[p++] = ht;
                } else if (ha >= 0.0f) {                                                /* :589-598 */
                    if (cfg->rewards && ha < cfg->agent_hit_validation[si]) hko_rw_hit(e, env, i, who);
                    o[p++] = ha;
                } else o[p++] = cfg->ray_distance[si];                                  /* :601 */
            }
        }
    }
}

int hko_get_observations(hko_env* e, float* obs)
{
    const int dim = hko_obs_dim(e);
    for (int env = 0; env < e->E; env++) hko_observe_env(e, env, obs + (size_t)env * e->A * dim);
    return 0;
}
