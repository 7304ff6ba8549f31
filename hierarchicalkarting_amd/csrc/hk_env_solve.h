// hk_env_solve.h — HierarchicalKartAgent.SolveLQR (HKA:699-1236) for every ego of a race instance: one wavefront
// per env, 16-lane group g = ego g.  Game assembly (players within 8 m, targets, the 7-branch heading heuristic with
// analytic wall raycasts, weights, reach-avoid costs) runs lane-per-player; the coupled Riccati solve is hk_lq_core.h.
#pragma once
#include "hk_env_device.h"
#include "hk_lq_core.h"

namespace hk {

struct KartL {                 // per-kart quantities staged in LDS for the assembly
    float px, pz, yaw, fx, fz, speed, heading, final_steer, msfs;
    int sec, straight;
    uint32_t flags;
    int pl1, pl2;              // plan_lane at (sec+1)%L, (sec+2)%L
    float pv1, pv2;
    float ray[5];              // nearest wall distance along sensors 0, 2, 4, 8, 6 (3e38 = none)
    float dC;                  // distance to the Trigger box of section (sec+1)%L
};

struct AsmGroup {              // per-ego assembly scratch (compact cost description)
    double QC[LQ_MAXP][4][LQ_MAXN];   // QC[i][b'][r] = Q_i[r][4b' + (r&3)]
    double QV[LQ_MAXP][LQ_MAXN];
    double tw[LQ_MAXP][4];
    double tgt[LQ_MAXP][4];
    double aw[LQ_MAXP][3];
    double opw[LQ_MAXP][3][3];
    double opt[LQ_MAXP][3][4];
    int M[LQ_MAXP];
    int pad_;
};

struct QCompact {
    const double* QC;
    const double* QV;
    __device__ double Q(int i, int r, int c) const { return ((c & 3) == (r & 3)) ? QC[(i * 4 + (c >> 2)) * LQ_MAXN + r] : 0.0; }
    __device__ double q(int i, int r) const { return QV[i * LQ_MAXN + r]; }
};

__device__ __forceinline__ double angle_difference(double a1, double a2)
{   // HKA:1341-1344
    return hk_atan2(hk_sin(a2 - a1), hk_cos(a2 - a1));
}

__global__ __launch_bounds__(64) void env_solve_kernel(EnvParams P, hk_agent_state* agents, const hk_env_state* envs,
                                                       hk_lq_debug* dbg_out, int* status)
{
    __shared__ LqGroupLds lds[4];
    __shared__ AsmGroup asg[4];
    __shared__ KartL kl[ENV_MAXA];
    const int env = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, r = lane & 15;
    const int A = P.A, L = P.L;
    const hk_env_state es = envs[env];
    const uint32_t all_mask = (1u << A) - 1u;
    if ((es.episode_steps % (A > 2 ? 4 : 1)) != 0) return;                       // HKA:317 cadence (Q9)
    if (!P.auto_reset && (es.inactive_mask & all_mask) == all_mask && (es.status & 4u)) return;
    hk_agent_state* ags = agents + (size_t)env * A;
    // ---- 1. stage karts
    if (lane < A) {
        const hk_agent_state* a = &ags[lane];
        KartL k;
        k.px = a->px; k.pz = a->pz; k.yaw = a->yaw;
        k.fx = hk_sinf(k.yaw); k.fz = hk_cosf(k.yaw);
        const float vx = a->vx, vz = a->vz;
        k.speed = mag3(vx, 0.0f, vz);
        float heading = hk_atan2f(k.fz, k.fx);                                    // HKA:734
        if (heading < 0) heading += TWO_PI_F;
        k.heading = heading;
        k.final_steer = a->final_steer;
        k.msfs = max_speed_for_state(P, k.yaw, vx, vz, a->wy, k.final_steer);
        k.sec = a->section_index;
        k.straight = is_straight(P, k.sec) ? 1 : 0;
        k.flags = a->flags;
        const int i1 = (k.sec + 1) % L, i2 = (k.sec + 2) % L;
        k.pl1 = a->plan_lane[i1]; k.pv1 = a->plan_vel[i1];
        k.pl2 = a->plan_lane[i2]; k.pv2 = a->plan_vel[i2];
        // BoxCollider.ClosestPoint distance to the next section's Trigger (HKA:846,876)
        {
            const SecDev& s = P.sec[i1];
            float relx = k.px - s.trig_x, relz = k.pz - s.trig_z;
            float lx = relx * s.fz + relz * (-s.fx);
            float lz = relx * s.fx + relz * s.fz;
            float dx = lx - f_clamp(lx, -TRIG_HX, TRIG_HX);
            float dz = lz - f_clamp(lz, -TRIG_HZ, TRIG_HZ);
            k.dC = sqrtf(dx * dx + dz * dz);
        }
        for (int q = 0; q < 5; q++) k.ray[q] = 3.0e38f;
        kl[lane] = k;
    }
    __syncthreads();
    // ---- 2. sensor rays of kart g against the candidate walls, 16 lanes wide
    if (g < A) {
        const int ssel[5] = {0, 2, 4, 8, 6};
        const KartL& k = kl[g];
        const float ox = k.px + SENSOR_LZ * k.fx, oz = k.pz + SENSOR_LZ * k.fz;
        float ddx[5], ddz[5], best[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            float ang = k.yaw + P.sensor_yaw[ssel[q]] * DEG2RAD_F;
            ddx[q] = hk_sinf(ang); ddz[q] = hk_cosf(ang);
            best[q] = 3.0e38f;
        }
        const int sidx = k.sec % L;
        const int w0 = P.far_off[sidx], w1 = P.far_off[sidx + 1];
        for (int w = w0 + r; w < w1; w += 16) {
            const hk_wall_seg ws = P.walls[P.far_idx[w]];
#pragma unroll
            for (int q = 0; q < 5; q++) {
                float t = ray_seg(ox, oz, ddx[q], ddz[q], ws);
                if (t >= 0.0f && t < best[q]) best[q] = t;
            }
        }
#pragma unroll
        for (int q = 0; q < 5; q++) {
            float b = best[q];
            b = f_min(b, __shfl_xor(b, 1, 64)); b = f_min(b, __shfl_xor(b, 2, 64));
            b = f_min(b, __shfl_xor(b, 4, 64)); b = f_min(b, __shfl_xor(b, 8, 64));
            if (r == 0) kl[g].ray[q] = b;
        }
    }
    __syncthreads();
    // ---- 3. game assembly: group g = ego g, lane r = player slot
    LqGroupLds& LG = lds[g];
    AsmGroup& AG = asg[g];
    const int ego = g;
    int N = 0, nearbyAgents = -1;
    int pl[ENV_MAXA] = {0, 0, 0, 0};
    bool solving = false;
    if (ego < A) {
        const uint32_t efl = kl[ego].flags;
        solving = (efl & HK_F_ENABLED) && P.low_mode[ego] == HK_LOW_LQR && !((es.inactive_mask >> ego) & 1u);
    }
    if (solving) {
        int all[ENV_MAXA], nall = 0;                                              // HKA:702
        all[nall++] = ego;
        for (int j = 0; j < P.n_team[ego]; j++) all[nall++] = P.team[ego][j];
        for (int j = 0; j < P.n_other[ego]; j++) all[nall++] = P.other[ego][j];
        if (A > 2) {                                                              // :709-720
            for (int q = 0; q < nall; q++) {
                const KartL& k = kl[all[q]];
                if (mag3(k.px - kl[ego].px, 0.0f, k.pz - kl[ego].pz) < 8) { nearbyAgents += 1; pl[N++] = all[q]; }
            }
        } else {
            for (int q = 0; q < nall; q++) pl[N++] = all[q];
        }
        nearbyAgents = nearbyAgents > 1 ? nearbyAgents : 1;                       // :725
    }
    const bool fixed = ego < A ? (P.high_mode[ego] == HK_HIGH_FIXED) : true;
    // zero the game inputs, then let the player lanes fill theirs
#pragma unroll
    for (int i = 0; i < LQ_MAXP; i++) {
        LG.Ab[i][r] = 0.0;
        if (r < 8) LG.Bb[i][r] = 0.0;
        if (r < 4) LG.Rb[i][r] = 0.0;
    }
    LG.x0[r] = 0.0;
    __syncthreads();
    int branch = 0;
    double controlcost = 0.0;
    if (solving && r < N) {
        const int i = r;
        const int ki = pl[i];
        const KartL& k = kl[ki];
        const KartL& me = kl[ego];
        const float dy = P.sec[0].marker_y - P.kart_y;                            // Q13
        const float speed = k.speed;
        double initial[4];
        initial[0] = k.px; initial[1] = k.pz; initial[2] = speed; initial[3] = k.heading;   // :731-736
        {   // LinearizedBicycle (KartLQRDynamics.cs:40-62), dt = Time.fixedDeltaTime widened to double (HKA:707)
            const double dt = (double)P.dt;
            double* Am = LG.Ab[i];
            double* Bm = LG.Bb[i];
            Am[0] = 1.0; Am[5] = 1.0; Am[10] = 1.0; Am[15] = 1.0;
            Am[0 * 4 + 2] = hk_cos(initial[3]) * dt;
            Am[1 * 4 + 2] = hk_sin(initial[3]) * dt;
            Am[0 * 4 + 3] = -hk_sin(initial[3]) * dt * initial[2];
            Am[1 * 4 + 3] = hk_cos(initial[3]) * dt * initial[2];
            Bm[2 * 2 + 0] = dt;
            Bm[3 * 2 + 1] = dt;
#pragma unroll
            for (int c = 0; c < 4; c++) LG.x0[4 * i + c] = initial[c];
        }
        const int s = k.sec + 1;                                                  // :746
        const int idx = s % L, idx2 = (s + 1) % L;
        int laneSel = 0, nextSel = 0;
        double vel = P.max_speed, nextVel = P.max_speed;
        if (ki == ego) {                                                          // :752-764, :782-794
            if (me.pl1 != 0) {
                laneSel = me.pl1;
                double pv = me.pv1 + (fixed ? 0 : P.vbucket[ego] * 2);
                vel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
            if (me.pl2 != 0) {
                nextSel = me.pl2;
                double pv = me.pv2 + (fixed ? 0 : P.vbucket[ego] * 2);
                nextVel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
        }   // else: the ego's belief about k's plan is only ever filled by the MCTS planner -> Trigger / max speed
        float lx, lz, nx, nz, cx, cz;
        lane_marker(P, idx, laneSel, lx, lz);
        lane_marker(P, idx2, nextSel, nx, nz);
        lane_marker(P, idx, 0, cx, cz);
        double target[4];
        target[0] = lx; target[1] = lz;
        target[2] = (speed <= 5.0f) ? 0.0f : vel;                                  // :810-817
        double fth;
        float targetHeading = hk_atan2f(lz - k.pz, lx - k.px);                     // :821
        if (targetHeading < 0) targetHeading += TWO_PI_F;
        if (mag3(lx - k.px, dy, lz - k.pz) <= (k.straight ? 10.5f : 7.5f)) {       // :823
            float h1 = hk_atan2f(lz - k.pz, lx - k.px);
            float h2 = hk_atan2f(nz - lz, nx - lx);
            float h5 = hk_atan2f(cz - k.pz, cx - k.px);
            float h6 = hk_atan2f(nz - k.pz, nx - k.px);
            const bool cutTrack = P.cut[(idx * 5 + laneSel) * 5 + nextSel] != 0;   // :832 (static geometry)
            const bool hit0 = k.ray[0] <= speed * 0.5f;                            // :834
            const bool hit1 = k.ray[1] <= 2.0f, hit2 = k.ray[2] <= 1.5f, hit3 = k.ray[3] <= 1.5f, hit4 = k.ray[4] <= 2.0f;
            const float dC = k.dC;
            const bool side = hit1 || hit2 || hit3 || hit4;
            if (cutTrack && dC > 4.0f) {                                           // B1 :846
                branch = 1;
                if (h5 < 0) h5 += TWO_PI_F;
                fth = h5;
                if (fth < 0) fth += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], fth);
            } else if ((side && (f_sign(h1) == f_sign(h5))) || hit0) {             // B2 :857 (Q12)
                branch = 2;
                if (h5 < 0) h5 += TWO_PI_F;
                fth = h5 - angle_difference(h1, h5) * 0.7f;
                if (fth < 0) fth += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], fth);
            } else if (side && (f_sign(h1) != f_sign(h5))) {                       // B3 :867
                branch = 3;
                if (h5 < 0) h5 += TWO_PI_F;
                fth = h5;
                if (fth < 0) fth += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], fth);
            } else if (dC <= 4.0f) {                                               // B4 :876
                branch = 4;
                target[0] = nx; target[1] = nz;
                if (speed > 5.0f) target[2] = nextVel;
                if (h6 < 0) h6 += TWO_PI_F;
                fth = h6;
                if (fth < 0) fth += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], fth);
            } else {                                                               // B5 :891
                branch = 5;
                if (h1 < 0) h1 += TWO_PI_F;
                if (h2 < 0) h2 += TWO_PI_F;
                fth = h1 - angle_difference(h2, h1) * 0.4f;
                if (fth < 0) fth += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], fth);
            }
        } else {
            const bool hit = k.ray[0] <= (k.straight ? 8.0f : 5.0f);               // :906
            if (hit) {                                                             // B6
                branch = 6;
                float h1 = hk_atan2f(cz - k.pz, cx - k.px);
                if (h1 < 0) h1 += TWO_PI_F;
                fth = initial[3] - angle_difference(initial[3], h1) * 0.85f;
            } else {                                                               // B7
                branch = 7;
                fth = initial[3] - angle_difference(initial[3], targetHeading);
            }
        }
        target[3] = fth;                                                           // :926
        double tw[4];                                                              // :930-964
        if (N > 2) tw[3] = (fixed ? 2.5 : 3.5) * nearbyAgents; else tw[3] = (fixed ? 1.9 : 3.5);
        if (speed <= 5.0f) {
            tw[0] = nearbyAgents * 0.3 * 3.1; tw[1] = nearbyAgents * 0.3 * 3.1; tw[2] = nearbyAgents * -2;
        } else {
            double mx = initial[2] > 1 ? initial[2] : 1;
            tw[0] = nearbyAgents * 0.3 * 3.1 / mx; tw[1] = nearbyAgents * 0.3 * 3.1 / mx; tw[2] = nearbyAgents * 5e-4;
        }
        float multiplier;                                                          // :976-1003
        if (A > 2 && N > 2) multiplier = (ki == ego ? (fixed ? 0.55f : 1.0f) : 1.7f) / nearbyAgents;
        else multiplier = (ki == ego ? (fixed ? 0.45f : 1.0f) : 1.3f);
        int M = 0, nearbyOpponents = 0;
        const int no = P.n_other[ki], nt = P.n_team[ki];
        for (int j = 0; j < no + nt; j++) {                                        // :1004-1190, k's own order (Q3)
            const bool isteam = j >= no;
            const int oi = isteam ? P.team[ki][j - no] : P.other[ki][j];
            bool member = false;
            for (int q = 0; q < N; q++) if (pl[q] == oi) member = true;
            if (!member) continue;
            const KartL& o = kl[oi];
            const float dist = mag3(o.px - k.px, 0.0f, o.pz - k.pz);
            const bool far = (dist > 8) || !(o.flags & HK_F_ACTIVE);
            double w = 0.0;
            if (!far) {
                float mult = isteam ? multiplier / 2.0f : multiplier;
                float pw = (float)((double)dist * sqrt((double)dist));            // Mathf.Pow(d, 1.5f)
                w = 1.0f / (pw * mult);
                if (!isteam) nearbyOpponents += 1;
            }
            AG.aw[i][M] = w;
            const int io = (o.sec + 1) % L;
            float olx, olz; double ov;
            if (oi == ego) {
                lane_marker(P, io, me.pl1, olx, olz);
                if (isteam) ov = me.msfs;
                else if (me.pl1 != 0) {
                    double pv = me.pv1 + (fixed ? 0 : P.vbucket[ego] * 2);
                    ov = (double)P.max_speed < pv ? (double)P.max_speed : pv;
                } else ov = P.max_speed;
            } else {
                lane_marker(P, io, 0, olx, olz);
                ov = isteam ? o.msfs : P.max_speed;
            }
            AG.opt[i][M][0] = olx; AG.opt[i][M][1] = olz; AG.opt[i][M][2] = ov; AG.opt[i][M][3] = 0.0;
            double mx = initial[2] > 1 ? initial[2] : 1;
            double w0, w1, w2;
            if (!isteam) {
                if (far) { w0 = 0.0; w1 = 0.0; w2 = 0; }
                else if (N > 2) { w0 = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents); w1 = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents); w2 = 0.08 / nearbyAgents; }
                else { w0 = (fixed ? 0.1 : 0.2) / mx; w1 = (fixed ? 0.1 : 0.2) / mx; w2 = 0.08; }
            } else {
                if (far || nearbyOpponents < 1) { w0 = 0.0; w1 = 0.0; w2 = 0; }
                else if (N > 2) { w0 = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents); w1 = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents); w2 = 0 / nearbyAgents; }
                else { w0 = -(fixed ? 1e-4 : 2e-4) / mx; w1 = -(fixed ? 1e-4 : 2e-4) / mx; w2 = 0; }
            }
            AG.opw[i][M][0] = w0; AG.opw[i][M][1] = w1; AG.opw[i][M][2] = w2;
            M++;
        }
        AG.M[i] = M;
        controlcost = 0.115;                                                       // :1192-1196
        if (N > 2) controlcost = fixed ? 0.135 : 0.25;
        LG.Rb[i][0] = 1.0 * controlcost; LG.Rb[i][3] = 1.0 * controlcost;          // getRMatrix KartLQRCosts.cs:132-140
#pragma unroll
        for (int c = 0; c < 4; c++) { AG.tw[i][c] = tw[c]; AG.tgt[i][c] = target[c]; }
        if (dbg_out && P.debug) {
            hk_lq_debug* d = &dbg_out[(size_t)env * A + ego];
            d->player_agent[i] = ki; d->branch[i] = branch; d->control_w[i] = controlcost;
            for (int c = 0; c < 4; c++) { d->initial[i][c] = initial[c]; d->target[i][c] = target[c]; d->target_w[i][c] = tw[c]; }
        }
    }
    __syncthreads();
    // ---- 3b. expand the compact reach-avoid cost (KartLQRCosts.cs:57-127) row by row: lane r = row r
    {
        const int b = r >> 2, sidx = r & 3;
        const int n = 4 * N;
#pragma unroll
        for (int i = 0; i < LQ_MAXP; i++) {
            double qc0 = 0.0, qc1 = 0.0, qc2 = 0.0, qc3 = 0.0, qv = 0.0;
            if (i < N && r < n) {
                const int M = AG.M[i];
                if (b == 0) {
                    double d = 0.0;
                    if (sidx < 2) {
                        double total = 0.0;                                        // :67-79
                        for (int j = 0; j < M; j++) total -= AG.aw[i][j];
                        d = total;
                    }
                    d += AG.tw[i][sidx];                                           // :81-84
                    qc0 = d;
                    if (sidx < 2) {
                        if (M > 0) qc1 = AG.aw[i][0];
                        if (M > 1) qc2 = AG.aw[i][1];
                        if (M > 2) qc3 = AG.aw[i][2];
                    }
                    double t = -AG.tgt[i][sidx];                                   // getQVec :109-113
                    qv = t * AG.tw[i][sidx];
                } else {
                    const int j = b - 1;
                    if (sidx < 2) qc0 = AG.aw[i][j];                               // :74
                    double dg = 0.0;
                    if (sidx < 3) dg = -AG.opw[i][j][sidx];                        // :91 assignment (Q4)
                    if (b == 1) qc1 = dg; else if (b == 2) qc2 = dg; else qc3 = dg;
                    qv = AG.opt[i][j][sidx];                                       // :117
                    if (sidx < 3) qv = qv * -AG.opw[i][j][sidx];                   // :121
                }
            }
            AG.QC[i][0][r] = qc0; AG.QC[i][1][r] = qc1; AG.QC[i][2][r] = qc2; AG.QC[i][3][r] = qc3;
            AG.QV[i][r] = qv;
        }
    }
    __syncthreads();
    // ---- 4. coupled Riccati solve
    int Nmax = N;
    Nmax = max(Nmax, __shfl_xor(Nmax, 16, 64));
    Nmax = max(Nmax, __shfl_xor(Nmax, 32, 64));
    Nmax = __builtin_amdgcn_readfirstlane(Nmax);
    if (Nmax == 0) return;
    QCompact qp;
    qp.QC = &AG.QC[0][0][0];
    qp.QV = &AG.QV[0][0];
    double u0[2];
    int singular = 0;
    lq_solve_group(r, N, Nmax, LG, qp, 3, u0, singular);                            // HKA:1201 horizon literal 3 (Q6)
    // ---- 5. decode (HKA:1206-1224)
    if (solving && r == 0) {
        hk_agent_state* me = &ags[ego];
        const float fs = kl[ego].final_steer;
        const float maxAng = fs * 0.4f;
        float angVel = f_clamp((float)u0[1], -maxAng, maxAng);
        uint32_t fl = me->flags;
        if (u0[0] < 0) { fl &= ~HK_F_ACCEL; fl |= HK_F_BRAKE; }
        else if (u0[0] > 0) { fl |= HK_F_ACCEL; fl &= ~HK_F_BRAKE; }
        else { fl &= ~(HK_F_ACCEL | HK_F_BRAKE); angVel = 0.0f; }                   // Q7
        me->flags = fl;
        me->steering = angVel / (0.4f * fs);
        if (singular) atomicOr(status, 1);
        if (dbg_out && P.debug) {
            hk_lq_debug* d = &dbg_out[(size_t)env * A + ego];
            d->n_players = N; d->u0[0] = u0[0]; d->u0[1] = u0[1];
        }
    }
}

}  // namespace hk
