/*
 * hk_oracle_policy.c — CPU ORACLE (test infrastructure): the ML-Agents PPO actor that drives LowMode == RL agents,
 * restated from the exported graphs under /root/reference/Assets/Karting/Prefabs/AI/ *.onnx (nodes listed by
 * tools/onnx_read.py: Sub, Div, Clip(+-5), Concat, {Gemm(transB), Sigmoid, Mul} x n, Gemm mu, Gemm branch, Exp, Softmax,
 * RandomNormalLike, Multinomial, Clip(+-3), Div 3, ArgMax) and the ML-Agents 2.0.1 runtime pieces around it:
 * StackingSensor (oldest observation first, zero-filled after a reset), DecisionRequester (DecisionPeriod, actions
 * repeated between decisions), KartAgent.OnActionReceived KA:440-448 -> HKA.InterpretDiscreteActions HKA:1371-1379.
 *
 * PARITY STATUS: mu / logits follow the graph exactly (fp32, every dot product a k-ascending fmaf chain seeded with the
 * bias — the order Barracuda uses is not documented, so agreement with Unity is to rounding, ~1e-6).  The sampled
 * outputs are "parity unpinned": Barracuda's RandomNormalLike / Multinomial stream cannot be reproduced outside Unity;
 * the draws here come from Philox-4x32 (counter = decision index and global agent row, key = seed and policy index).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hk_oracle.h"
#include "../include/hk_detmath.h"
#include "hk_oracle_internal.h"

struct hko_policy {
    hk_policy_desc d;          /* pointers below are owned copies */
    float* mem;
    int n_slots;
    int slots[HK_MAX_AGENTS];
    float* ring;               /* [E][n_slots][stack][obs_dim] */
    int32_t* epoch;            /* [E]: episodes_done + initial_started the ring belongs to; -1 = invalid */
};

static const float* dup(float** cursor, const float* src, size_t n)
{
    float* dst = *cursor;
    memcpy(dst, src, n * sizeof(float));
    *cursor += n;
    return dst;
}

static int desc_ok(const hk_policy_desc* d)
{
    if (!d || d->in_dim < 2 || d->in_dim > HK_POLICY_MAX_IN || (d->in_dim & 1) || d->stack < 1 || d->in_dim % d->stack) return 0;
    if (d->hidden < 32 || d->hidden > HK_POLICY_MAX_HIDDEN || d->hidden % 32) return 0;
    if (d->n_layers < 1 || d->n_layers > HK_POLICY_MAX_LAYERS || d->n_branch < 1 || d->n_branch > 7) return 0;
    if (d->normalize && (!d->norm_mean || !d->norm_std)) return 0;
    for (int l = 0; l < d->n_layers; l++) if (!d->W[l] || !d->b[l]) return 0;
    return d->W_mu && d->b_mu && d->log_sigma && d->W_branch && d->b_branch;
}

int hko_policy_attach(hko_env* e, const hk_policy_desc* d, const int32_t* agent_slots, int n_slots, int decision_period)
{
    if (!e || !desc_ok(d) || n_slots < 1 || n_slots > e->A || decision_period < 1) return HK_ERR_INVALID;
    if (e->n_policies >= HK_MAX_POLICIES) return HK_ERR_INVALID;
    if (d->in_dim != hko_obs_dim(e) * d->stack) return HK_ERR_INVALID;
    for (int j = 0; j < n_slots; j++) {
        if (agent_slots[j] < 0 || agent_slots[j] >= e->A || e->cfg.low_mode[agent_slots[j]] != HK_LOW_RL) return HK_ERR_INVALID;
        for (int p = 0; p < e->n_policies; p++)
            for (int q = 0; q < e->policy[p]->n_slots; q++)
                if (e->policy[p]->slots[q] == agent_slots[j]) return HK_ERR_INVALID;
    }
    struct hko_policy* P = (struct hko_policy*)calloc(1, sizeof(*P));
    P->d = *d;
    const int H = d->hidden, K0 = d->in_dim;
    size_t total = 2 * (size_t)K0 + (size_t)H * K0 + (size_t)(d->n_layers - 1) * H * H + (size_t)d->n_layers * H + H + 2 + (size_t)d->n_branch * (H + 1);
    P->mem = (float*)malloc(total * sizeof(float));
    float* c = P->mem;
    if (d->normalize) { P->d.norm_mean = dup(&c, d->norm_mean, K0); P->d.norm_std = dup(&c, d->norm_std, K0); }
    for (int l = 0; l < d->n_layers; l++) {
        P->d.W[l] = dup(&c, d->W[l], (size_t)H * (l == 0 ? K0 : H));
        P->d.b[l] = dup(&c, d->b[l], H);
    }
    P->d.W_mu = dup(&c, d->W_mu, H); P->d.b_mu = dup(&c, d->b_mu, 1); P->d.log_sigma = dup(&c, d->log_sigma, 1);
    P->d.W_branch = dup(&c, d->W_branch, (size_t)d->n_branch * H); P->d.b_branch = dup(&c, d->b_branch, d->n_branch);
    P->n_slots = n_slots;
    for (int j = 0; j < n_slots; j++) P->slots[j] = agent_slots[j];
    P->ring = (float*)calloc((size_t)e->E * n_slots * K0, sizeof(float));
    P->epoch = (int32_t*)malloc(sizeof(int32_t) * e->E);
    for (int env = 0; env < e->E; env++) P->epoch[env] = -1;
    if (!e->obs_scratch) e->obs_scratch = (float*)malloc(sizeof(float) * (size_t)e->E * e->A * hko_obs_dim(e));
    e->decision_period = decision_period;
    e->policy[e->n_policies] = P;
    return e->n_policies++;
}

void hko_policy_free(hko_env* e)
{
    for (int p = 0; p < e->n_policies; p++) {
        free(e->policy[p]->mem); free(e->policy[p]->ring); free(e->policy[p]->epoch); free(e->policy[p]);
    }
    free(e->obs_scratch);
    e->n_policies = 0;
}

void hko_policy_invalidate(hko_env* e, int env)
{
    for (int p = 0; p < e->n_policies; p++) e->policy[p]->epoch[env] = -1;
}

static inline float swish(float s) { return hk_swishf(s); }     /* Sigmoid then Mul in the graph */

/* x: one stacked observation row [in_dim] (oldest first).  mu[1], logits[n_branch] */
static void forward_row(const hk_policy_desc* d, const float* x, float* mu, float* logits)
{
    float a[HK_POLICY_MAX_IN], y[HK_POLICY_MAX_HIDDEN];
    int K = d->in_dim;
    for (int k = 0; k < K; k++) {
        float v = x[k];
        if (d->normalize) {
            v = (v - d->norm_mean[k]) / d->norm_std[k];
            v = v < -5.0f ? -5.0f : (v > 5.0f ? 5.0f : v);
        }
        a[k] = v;
    }
    const int H = d->hidden;
    for (int l = 0; l < d->n_layers; l++) {
        for (int j = 0; j < H; j++) {
            float s = d->b[l][j];
            const float* w = d->W[l] + (size_t)j * K;
            for (int k = 0; k < K; k++) s = fmaf(a[k], w[k], s);
            y[j] = swish(s);
        }
        for (int j = 0; j < H; j++) a[j] = y[j];
        K = H;
    }
    float s = d->b_mu[0];
    for (int k = 0; k < H; k++) s = fmaf(a[k], d->W_mu[k], s);
    *mu = s;
    for (int b = 0; b < d->n_branch; b++) {
        s = d->b_branch[b];
        for (int k = 0; k < H; k++) s = fmaf(a[k], d->W_branch[(size_t)b * H + k], s);
        logits[b] = s;
    }
}

int hko_policy_forward(hko_env* e, int policy, int rows, const float* obs, float* mu, float* logits)
{
    if (!e || policy < 0 || policy >= e->n_policies || rows < 0) return HK_ERR_INVALID;
    const hk_policy_desc* d = &e->policy[policy]->d;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) forward_row(d, obs + (size_t)r * d->in_dim, &mu[r], &logits[(size_t)r * d->n_branch]);
    return 0;
}

/* continuous_actions / discrete_actions (or their deterministic_ twins) of one row */
static void sample_row(const hk_policy_desc* d, int pidx, uint64_t decision, uint32_t row, float mu, const float* logits,
                       float* steer, int32_t* branch)
{
    uint32_t r[4];
    philox4x32((uint32_t)decision, (uint32_t)(decision >> 32), row, 0x504F4C49u, d->seed, (uint32_t)pidx, r);
    float eps = 0.0f;
    if (!d->deterministic) {
        float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f);      /* (0, 1] */
        float u2 = u01(r[1]);
        eps = sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);  /* Box-Muller */
    }
    float sigma = hk_expf(d->log_sigma[0]);
    float v = mu + eps * sigma;
    v = v < -3.0f ? -3.0f : (v > 3.0f ? 3.0f : v);
    *steer = v / 3.0f;
    int best = 0;
    for (int b = 1; b < d->n_branch; b++) if (logits[b] > logits[best]) best = b;   /* ArgMax: first maximum */
    if (d->deterministic) { *branch = best; return; }
    float ex[8], tot = 0.0f;
    for (int b = 0; b < d->n_branch; b++) { ex[b] = hk_expf(logits[b] - logits[best]); tot += ex[b]; }
    float thr = u01(r[2]) * tot, cum = 0.0f;
    int pick = d->n_branch - 1;
    for (int b = 0; b < d->n_branch; b++) { cum += ex[b]; if (thr < cum) { pick = b; break; } }
    *branch = pick;
}

void hko_policy_decide(hko_env* e)
{
    if (e->n_policies == 0 || (e->academy_step % e->decision_period) != 0) return;
    const uint64_t decision = (uint64_t)(e->academy_step / e->decision_period);
    const int dim = hko_obs_dim(e), A = e->A;
#pragma omp parallel for schedule(dynamic, 8)
    for (int env = 0; env < e->E; env++) {
        float* obs = e->obs_scratch + (size_t)env * A * dim;
        hko_observe_env(e, env, obs);
        const int32_t ep = e->es[env].episodes_done + e->es[env].initial_started;
        for (int p = 0; p < e->n_policies; p++) {
            struct hko_policy* P = e->policy[p];
            const hk_policy_desc* d = &P->d;
            const int S = d->stack, w = (int)(decision % (uint64_t)S);
            float* ring = P->ring + (size_t)env * P->n_slots * d->in_dim;
            if (P->epoch[env] != ep) {                                   /* Agent reset -> StackingSensor.Reset: zeros */
                memset(ring, 0, sizeof(float) * P->n_slots * d->in_dim);
                P->epoch[env] = ep;
            }
            for (int j = 0; j < P->n_slots; j++) {
                const int agent = P->slots[j];
                float* rj = ring + (size_t)j * d->in_dim;
                memcpy(rj + (size_t)w * dim, obs + (size_t)agent * dim, sizeof(float) * dim);
                float x[HK_POLICY_MAX_IN];
                for (int i = 0; i < S; i++) memcpy(x + (size_t)i * dim, rj + (size_t)((w + 1 + i) % S) * dim, sizeof(float) * dim);
                float mu, logits[8];
                forward_row(d, x, &mu, logits);
                const uint32_t row = (uint32_t)(e->cfg.env_id_base + env) * (uint32_t)A + (uint32_t)agent;
                sample_row(d, p, decision, row, mu, logits, &e->act_steer[(size_t)env * A + agent], &e->act_branch[(size_t)env * A + agent]);
            }
        }
    }
}

int hko_get_actions(hko_env* e, float* steer, int32_t* branch)
{
    size_t na = (size_t)e->E * e->A;
    memcpy(steer, e->act_steer, na * sizeof(float));
    memcpy(branch, e->act_branch, na * sizeof(int32_t));
    return 0;
}
