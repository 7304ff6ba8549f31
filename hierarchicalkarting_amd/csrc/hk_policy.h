// hk_policy.h — on-device inference of the ML-Agents PPO actor that drives LowMode == RL agents (contract
// in include/hk.h).  Three steps per decision tick, all on the handle's stream:
//   env_observe_kernel      CollectObservations of every agent                      (hk_env_observe.h)
//   policy_stack_kernel     StackingSensor: push the newest observation into the agent's ring, zero the ring first when
//                           the episode changed since the last decision
//   policy_mlp_kernel       normalise -> n x (Linear + Swish) -> {mu, logits} -> sample -> latch steer / branch
//
// policy_mlp_kernel: one workgroup (8 waves) per tile of 64 rows (row = one agent's stacked observation).  The tile's
// activations live TRANSPOSED in LDS (At[k][row], row stride 65 floats: conflict-free for the column writes of the loader
// and of the layer epilogue, and for the 32-consecutive-row reads of the MFMA A operand) and are overwritten in place
// layer by layer (the first layer streams its inputs through in chunks of <= 320, so stacks of 8 observations fit); weights are pre-transposed once at attach time (Wt[k][out]) so the MFMA B operand is a coalesced
// 128-byte read per half-wave straight from L2 (all layers together are <= 0.6 MB).  At 66.5 KB of LDS and <= 128 VGPRs
// two workgroups share a CU, so one's loader / Swish epilogue / barriers overlap the other's matrix work.  The GEMMs run on the f32-input
// matrix instruction v_mfma_f32_32x32x2_f32: a 64 x H layer is (H/32) x 2 blocks of 32 x 32, two per wave at H = 256.
// Its result is bit-for-bit a k-ascending fmaf chain seeded with the accumulator input, which we seed with the bias —
// exactly the chain the CPU oracle evaluates, so mu / logits agree bit for bit.  The two heads (1 + n_branch outputs)
// are plain fmaf chains on the vector ALU (N = 4 is not MFMA-shaped).
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/hk.h"
#include "../../include/hk_detmath.h"
#include "hk_swish.h"      // Sigmoid then Mul in the exported graph
#include "hk_env_device.h"

namespace hk {

constexpr int PM_TILE = 64;                 // rows per workgroup
constexpr int PM_LD = PM_TILE + 1;          // LDS row stride of At (floats)
constexpr int PM_THREADS = 512;
constexpr int PM_MAX_OUT = 8;               // 1 + n_branch
constexpr int PM_TRIP = 4;                  // k-steps per operand set of the gemm's ping-pong (8 spills at 128 VGPRs)
constexpr int PM_KC_MAX = 256;              // 256 x 65 floats = 66.5 KB of LDS: two workgroups per CU

struct PolicyParams {
    int in_dim, obs_dim, stack, hidden, n_layers, n_branch, normalize, deterministic;
    int kc;                     // layer-0 inputs staged through LDS per chunk (even, <= PM_KC_MAX); later layers use hidden
    uint32_t seed;
    int index;                  // policy index (Philox key word)
    int n_slots;
    int slots[HK_MAX_AGENTS];
    const float* mean;
    const float* std;
    const float* Wt[HK_POLICY_MAX_LAYERS];   // [k][hidden]
    const float4* Wq[HK_POLICY_MAX_LAYERS];  // [k / 8][2][hidden] x {4 k-steps}: element (g, half, col)[j] = W[k = 8 g + 2 j + half][col] — the B operands of four consecutive
                                             // MFMA k-steps of one lane as ONE 16-byte load (whole groups of 8 inputs only; the remainder reads Wt)
    const float* b[HK_POLICY_MAX_LAYERS];
    const float* W_mu;                       // [hidden]
    const float* b_mu;
    const float* log_sigma;
    const float* W_branch;                   // [n_branch][hidden]
    const float* b_branch;
    float* ring;                             // [E][n_slots][stack][obs_dim]
    int32_t* epoch;                          // [E][n_slots]
};

struct PolicyDevice {
    PolicyParams q{};
    float* weights = nullptr;                // one allocation behind every const float* above
    bool used = false;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------
// StackingSensor.  One wave per (env, slot), four to a block; w = ring slot that receives the newest observation.  (Round 5: a wave needs no block
// barrier — the epoch word is read by every lane before lane 0 rewrites it, the zero fill precedes the copy in program order — and a pair moves
// obs_dim <= 126 floats: with 128 threads and two __syncthreads per pair the kernel took 1.9 ms per decision beside the planner's searches.)
__global__ __launch_bounds__(256) void policy_stack_kernel(PolicyParams Q, int E, int A, const hk_env_state* envs_by_slot, const int* slot_of,
                                                           const float* obs, int w)
{
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t = threadIdx.x & 63;
    if (pair >= E * Q.n_slots) return;                         // (wave-uniform)
    const int env = pair / Q.n_slots, j = pair % Q.n_slots;
    const hk_env_state* es = &envs_by_slot[slot_of[env]];      // env words are stored by lane-group slot (hk_env_device.h)
    const int ep = es->episodes_done + es->initial_started;
    const bool stale = Q.epoch[pair] != ep;                    // every lane reads the word; lane 0 rewrites it below, after this load in program order
    float* ring = Q.ring + (size_t)pair * Q.in_dim;
    if (stale) {
        for (int k = t; k < Q.in_dim; k += 64) ring[k] = 0.0f;
        if (t == 0) Q.epoch[pair] = ep;
    }
    // (the copy of slot w lands on addresses the zero fill may have written from other lanes of this wave: a wave's stores reach memory in program order)
    const float* o = obs + ((size_t)env * A + Q.slots[j]) * Q.obs_dim;
    for (int k = t; k < Q.obs_dim; k += 64) ring[(size_t)w * Q.obs_dim + k] = o[k];
}

// hk_reset: force the rings of the listed envs (all when env_ids == nullptr) to be cleared at the next decision
__global__ __launch_bounds__(256) void policy_invalidate_kernel(PolicyParams Q, const int* env_ids, int n)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * Q.n_slots) return;
    const int env = env_ids ? env_ids[idx / Q.n_slots] : idx / Q.n_slots;
    Q.epoch[(size_t)env * Q.n_slots + idx % Q.n_slots] = -1;
}


// One wave's share of a layer over kc inputs: acc0 (+ acc1 when HAS1) += A[32 rows][kc] * B[kc][32 cols], as MFMA
// 32x32x2 f32 steps in ascending k.  a?p already point at this lane's first element of A (LDS, row stride PM_LD per k); B comes from
// global memory (L2): whole groups of 8 inputs (4 k-steps) through q?p — this lane's float4 of the group-major copy Wq, 2 * H float4 per
// group — the remainder (kc % 8 inputs) through b?p (Wt, row stride H per k), one k-step at a time.  SAMEB: both blocks use the same B
// columns (H = 256).
// Pipeline (round 5): the B loads run THREE groups ahead of the MFMAs that use them (four register sets), the A reads one group ahead
// (two sets).  With the loads one group ahead only — 8 MFMAs = 512 matrix-pipe cycles, ~2 000 with four waves taking turns — a weight
// load's ~1.5 us from L2 under load was not covered and the pipe sat at 62 % (SQ_VALU_MFMA_BUSY_CYCLES) whatever the other workgroup
// of the CU was doing (a start stagger of the two workgroups of a CU changed nothing — f32 MFMAs and vector instructions do not
// co-execute on gfx950, hk_swish.h — see profiles/r05_i_actor.txt).
template <bool HAS1, bool SAMEB>
__device__ __forceinline__ void pm_gemm(f32x16& acc0, f32x16& acc1, const float* a0p, const float* a1p,
                                        const float* __restrict__ b0p, const float* __restrict__ b1p,
                                        const float4* __restrict__ q0p, const float4* __restrict__ q1p, int kc, int H)
{
    const int ng = kc >> 3;
    const size_t qs = (size_t)2 * H;                     // float4 per group
    auto ldB = [&](int g, float4& x0, float4& x1) {
        x0 = q0p[(size_t)g * qs];
        if (HAS1 && !SAMEB) x1 = q1p[(size_t)g * qs];
    };
    auto ldA = [&](int g, float (&x0)[4], float (&x1)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x0[j] = a0p[(size_t)(8 * g + 2 * j) * PM_LD];
            if (HAS1) x1[j] = a1p[(size_t)(8 * g + 2 * j) * PM_LD];
        }
    };
    auto mm = [&](const float4& y0, const float4& y1, const float (&x0)[4], const float (&x1)[4]) {
        const float b0v[4] = {y0.x, y0.y, y0.z, y0.w};
        const float b1v[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[j], b0v[j], acc0, 0, 0, 0);
            if (HAS1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[j], SAMEB ? b0v[j] : b1v[j], acc1, 0, 0, 0);
        }
    };
    if (ng > 0) {
        float4 B0a, B0b, B0c, B0d, B1a = {}, B1b = {}, B1c = {}, B1d = {};
        float A0p[4], A1p[4] = {}, A0q[4], A1q[4] = {};
        ldB(0, B0a, B1a);
        if (ng > 1) ldB(1, B0b, B1b);
        if (ng > 2) ldB(2, B0c, B1c);
        ldA(0, A0p, A1p);
        for (int g = 0; g < ng; g += 4) {
            if (g + 3 < ng) ldB(g + 3, B0d, B1d);
            if (g + 1 < ng) ldA(g + 1, A0q, A1q);
            mm(B0a, B1a, A0p, A1p);
            if (g + 1 >= ng) break;
            if (g + 4 < ng) ldB(g + 4, B0a, B1a);
            if (g + 2 < ng) ldA(g + 2, A0p, A1p);
            mm(B0b, B1b, A0q, A1q);
            if (g + 2 >= ng) break;
            if (g + 5 < ng) ldB(g + 5, B0b, B1b);
            if (g + 3 < ng) ldA(g + 3, A0q, A1q);
            mm(B0c, B1c, A0p, A1p);
            if (g + 3 >= ng) break;
            if (g + 6 < ng) ldB(g + 6, B0c, B1c);
            if (g + 4 < ng) ldA(g + 4, A0p, A1p);
            mm(B0d, B1d, A0q, A1q);
        }
    }
    for (int k0 = ng * 8; k0 < kc; k0 += 2) {
        const float bv0 = b0p[(size_t)k0 * H];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[(size_t)k0 * PM_LD], bv0, acc0, 0, 0, 0);
        if (HAS1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[(size_t)k0 * PM_LD], SAMEB ? bv0 : b1p[(size_t)k0 * H], acc1, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// src: [rows][in_dim]; logical element k of a row lives at ((w + 1 + k / obs_dim) % stack) * obs_dim + k % obs_dim
// (w = stack - 1 for plain oldest-first rows).  Outputs: mu_out / logit_out when non-null; act_steer / act_branch
// (indexed [env][agent], row = env * n_slots + j) when non-null.
// MODE: how the (H / 32) x 2 blocks of a layer fall on the 8 waves — 0: one block per wave (H <= 128), 1: two blocks per
// wave sharing their B columns (H = 256), 2: mixed (other H), decided per wave — three inlined forms of the gemm: held to two waves per SIMD
// instead of four (at 128 registers it spills 85), i.e. one workgroup per CU.
template <int MODE>
__global__ __launch_bounds__(PM_THREADS, MODE == 2 ? 2 : 4) void policy_mlp_kernel(PolicyParams Q, int rows, const float* src, int w,
                                                                   unsigned long long decision, int env_id_base, int A,
                                                                   float* mu_out, float* logit_out, float* act_steer, int* act_branch)
{
    extern __shared__ __align__(16) float At[];        // [max(kc, hidden)][PM_LD] then head[PM_MAX_OUT][PM_TILE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: block / unit selection stays scalar
    const int row0 = blockIdx.x * PM_TILE;
    const int K0 = Q.in_dim, H = Q.hidden, KC = Q.kc;
    float* head = At + (size_t)(KC > H ? KC : H) * PM_LD;

    // Blocks of 32 x 32 of a 64 x H layer: cb = column block, rb = row block (0, 1); <= 16 blocks, two per wave at most.
    const int ncb = H >> 5;
    const int nunits = ncb * 2;
    const int half = lane >> 5, c = lane & 31;
    const int u0 = wave, u1 = wave + 8;
    const bool has0 = u0 < nunits, has1 = MODE == 0 ? false : (MODE == 1 ? true : u1 < nunits);
    const int cb0 = u0 % ncb, rb0 = u0 / ncb;
    const int cb1 = has1 ? u1 % ncb : cb0, rb1 = has1 ? u1 / ncb : rb0;
    const float* a0p = At + (size_t)half * PM_LD + rb0 * 32 + c;
    const float* a1p = At + (size_t)half * PM_LD + rb1 * 32 + c;

    for (int l = 0; l < Q.n_layers; l++) {
        const int K = l == 0 ? K0 : H;
        f32x16 acc0, acc1;
        {
            const float bias0 = has0 ? Q.b[l][cb0 * 32 + c] : 0.0f;
            const float bias1 = has1 ? Q.b[l][cb1 * 32 + c] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = bias0; acc1[r] = bias1; }
        }
        // layer 0 streams its K0 inputs through LDS in chunks of KC (the accumulators carry the fmaf chain across chunks);
        // the later layers read the activations the previous epilogue left in LDS
        for (int kb = 0; kb < K; kb += KC) {
            const int kc = (K - kb) < KC ? (K - kb) : KC;
            if (l == 0) {
                if (kb > 0) __syncthreads();                      // the previous chunk has been consumed
                // ---- load + normalise the tile chunk, transposed.  Lanes run over k (coalesced reads, conflict-free
                // LDS column writes), waves over rows; the per-k constants (ring offset, mean, std) are set up once.
                constexpr int MMAX = (PM_KC_MAX + 63) / 64;
                int soff[MMAX];
                float mean[MMAX], sdev[MMAX];
#pragma unroll
                for (int m = 0; m < MMAX; m++) {
                    const int kl = lane + 64 * m, k = kb + kl;
                    soff[m] = -1; mean[m] = 0.0f; sdev[m] = 1.0f;
                    if (kl < kc) {
                        const int i = k / Q.obs_dim, kk = k - i * Q.obs_dim;
                        int slot = w + 1 + i; slot -= (slot >= Q.stack) ? Q.stack : 0;
                        soff[m] = slot * Q.obs_dim + kk;
                        if (Q.normalize) { mean[m] = Q.mean[k]; sdev[m] = Q.std[k]; }
                    }
                }
                for (int r = wave; r < PM_TILE; r += PM_THREADS / 64) {
                    const bool rowok = row0 + r < rows;
                    const float* srow = src + (size_t)(row0 + r) * K0;
                    float v[MMAX];
#pragma unroll
                    for (int m = 0; m < MMAX; m++) v[m] = (rowok && soff[m] >= 0) ? srow[soff[m]] : 0.0f;
#pragma unroll
                    for (int m = 0; m < MMAX; m++) {
                        if (soff[m] < 0) continue;
                        float x = v[m];
                        if (Q.normalize && rowok) {
                            x = (x - mean[m]) / sdev[m];
                            x = x < -5.0f ? -5.0f : (x > 5.0f ? 5.0f : x);
                        }
                        At[(size_t)(lane + 64 * m) * PM_LD + r] = x;
                    }
                }
                __syncthreads();
            }
            if (has0) {
                const float* __restrict__ Wt = Q.Wt[l] + (size_t)kb * H;
                const float* b0p = Wt + (size_t)half * H + cb0 * 32 + c;
                const float* b1p = Wt + (size_t)half * H + cb1 * 32 + c;
                const float4* __restrict__ Wq = Q.Wq[l] + (size_t)(kb >> 3) * 2 * H;          // (kb is a multiple of 8: policy_upload)
                const float4* q0p = Wq + (size_t)half * H + cb0 * 32 + c;
                const float4* q1p = Wq + (size_t)half * H + cb1 * 32 + c;
                if (MODE == 0) pm_gemm<false, false>(acc0, acc1, a0p, a1p, b0p, b1p, q0p, q1p, kc, H);
                else if (MODE == 1) pm_gemm<true, true>(acc0, acc1, a0p, a1p, b0p, b1p, q0p, q1p, kc, H);
                else if (!has1) pm_gemm<false, false>(acc0, acc1, a0p, a1p, b0p, b1p, q0p, q1p, kc, H);
                else if (cb1 == cb0) pm_gemm<true, true>(acc0, acc1, a0p, a1p, b0p, b1p, q0p, q1p, kc, H);
                else pm_gemm<true, false>(acc0, acc1, a0p, a1p, b0p, b1p, q0p, q1p, kc, H);
            }
        }
        if (has0) {
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = swish(acc0[r]); if (MODE != 0 && has1) acc1[r] = swish(acc1[r]); }
        }
        __syncthreads();            // every wave has finished reading this layer's input
        if (has0) {
            // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                At[(size_t)(cb0 * 32 + c) * PM_LD + rb0 * 32 + rr] = acc0[r];
                if (MODE != 0 && has1) At[(size_t)(cb1 * 32 + c) * PM_LD + rb1 * 32 + rr] = acc1[r];
            }
        }
        __syncthreads();
    }

    // ---- heads: out 0 = mu, 1.. = branch logits; thread -> (out = tid / 64, row = tid % 64)
    {
        const int out = tid >> 6, r = tid & 63;
        if (out < 1 + Q.n_branch) {
            const float* wv = out == 0 ? Q.W_mu : Q.W_branch + (size_t)(out - 1) * H;
            float s = out == 0 ? Q.b_mu[0] : Q.b_branch[out - 1];
            for (int k = 0; k < H; k++) s = __builtin_fmaf(At[(size_t)k * PM_LD + r], wv[k], s);
            head[out * PM_TILE + r] = s;
        }
    }
    __syncthreads();
    if (tid < PM_TILE && row0 + tid < rows) {
        const int row = row0 + tid;
        const float mu = head[tid];
        float lg[PM_MAX_OUT];
        for (int b = 0; b < Q.n_branch; b++) lg[b] = head[(1 + b) * PM_TILE + tid];
        if (mu_out) mu_out[row] = mu;
        if (logit_out) for (int b = 0; b < Q.n_branch; b++) logit_out[(size_t)row * Q.n_branch + b] = lg[b];
        if (act_steer) {
            const int env = row / Q.n_slots, agent = Q.slots[row % Q.n_slots];
            const uint32_t grow = (uint32_t)(env_id_base + env) * (uint32_t)A + (uint32_t)agent;
            uint32_t rnd[4];
            philox4x32((uint32_t)decision, (uint32_t)(decision >> 32), grow, 0x504F4C49u, Q.seed, (uint32_t)Q.index, rnd);
            float eps = 0.0f;
            if (!Q.deterministic) {
                const float u1 = (float)((rnd[0] >> 8) + 1u) * (1.0f / 16777216.0f);
                const float u2 = u01(rnd[1]);
                eps = sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);
            }
            const float sigma = hk_expf(Q.log_sigma[0]);
            float v = mu + eps * sigma;
            v = v < -3.0f ? -3.0f : (v > 3.0f ? 3.0f : v);
            int best = 0;
            for (int b = 1; b < Q.n_branch; b++) if (lg[b] > lg[best]) best = b;
            int pick = best;
            if (!Q.deterministic) {
                float ex[PM_MAX_OUT], tot = 0.0f;
                for (int b = 0; b < Q.n_branch; b++) { ex[b] = hk_expf(lg[b] - lg[best]); tot += ex[b]; }
                const float thr = u01(rnd[2]) * tot;
                float cum = 0.0f;
                pick = Q.n_branch - 1;
                for (int b = 0; b < Q.n_branch; b++) { cum += ex[b]; if (thr < cum) { pick = b; break; } }
            }
            act_steer[(size_t)env * A + agent] = v / 3.0f;
            act_branch[(size_t)env * A + agent] = pick;
        }
    }
}

inline size_t policy_lds_bytes(const PolicyParams& q)
{
    const int kmax = q.kc > q.hidden ? q.kc : q.hidden;
    return ((size_t)kmax * PM_LD + (size_t)PM_MAX_OUT * PM_TILE) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------------
inline int policy_validate(const hk_policy_desc* d, std::string& err)
{
    if (!d) { err = "hk_policy_attach: NULL desc"; return HK_ERR_INVALID; }
    if (d->in_dim < 2 || (d->in_dim & 1) || d->stack < 1 || d->in_dim % d->stack) { err = "hk_policy_attach: bad in_dim / stack"; return HK_ERR_INVALID; }
    if (d->in_dim > HK_POLICY_MAX_IN) { err = "hk_policy_attach: in_dim > HK_POLICY_MAX_IN"; return HK_ERR_UNSUPPORTED; }
    if (d->hidden < 32 || d->hidden > HK_POLICY_MAX_HIDDEN || d->hidden % 32) { err = "hk_policy_attach: hidden must be a multiple of 32 in [32, 256]"; return HK_ERR_INVALID; }
    if (d->n_layers < 1 || d->n_layers > HK_POLICY_MAX_LAYERS || d->n_branch < 1 || d->n_branch > PM_MAX_OUT - 1) { err = "hk_policy_attach: bad n_layers / n_branch"; return HK_ERR_INVALID; }
    if (d->normalize && (!d->norm_mean || !d->norm_std)) { err = "hk_policy_attach: normalize without mean / std"; return HK_ERR_INVALID; }
    for (int l = 0; l < d->n_layers; l++) if (!d->W[l] || !d->b[l]) { err = "hk_policy_attach: NULL layer weights"; return HK_ERR_INVALID; }
    if (!d->W_mu || !d->b_mu || !d->log_sigma || !d->W_branch || !d->b_branch) { err = "hk_policy_attach: NULL head weights"; return HK_ERR_INVALID; }
    return HK_OK;
}

// upload (weights of the hidden layers transposed to [k][hidden]); E == 0: no ring (a forward-only policy)
inline int policy_upload(PolicyDevice& pd, const hk_policy_desc* d, int index, int obs_dim, const int32_t* slots, int n_slots,
                         int E, hipStream_t stream, std::string& err)
{
    const int H = d->hidden, K0 = d->in_dim;
    std::vector<float> host;
    auto put = [&](const float* src, size_t n) { size_t off = host.size(); host.insert(host.end(), src, src + n); return off; };
    size_t o_mean = 0, o_std = 0, o_W[HK_POLICY_MAX_LAYERS], o_b[HK_POLICY_MAX_LAYERS], o_Wq[HK_POLICY_MAX_LAYERS];
    while (host.size() % 4) host.push_back(0.0f);        // (everything below keeps the float4 copies 16-byte aligned: see o_Wq)
    if (d->normalize) { o_mean = put(d->norm_mean, K0); o_std = put(d->norm_std, K0); }
    for (int l = 0; l < d->n_layers; l++) {
        const int K = l == 0 ? K0 : H;
        o_W[l] = host.size();
        host.resize(host.size() + (size_t)K * H);
        float* wt = host.data() + o_W[l];
        for (int j = 0; j < H; j++)
            for (int k = 0; k < K; k++) wt[(size_t)k * H + j] = d->W[l][(size_t)j * K + k];
        o_b[l] = put(d->b[l], H);
        // the group-major copy (PolicyParams::Wq): whole groups of 8 inputs
        while (host.size() % 4) host.push_back(0.0f);
        o_Wq[l] = host.size();
        const int ngr = K / 8;
        host.resize(host.size() + (size_t)ngr * 2 * H * 4);
        float* wq = host.data() + o_Wq[l];
        const float* wt2 = host.data() + o_W[l];
        for (int g = 0; g < ngr; g++)
            for (int hf = 0; hf < 2; hf++)
                for (int j = 0; j < H; j++)
                    for (int q4 = 0; q4 < 4; q4++) wq[(((size_t)g * 2 + hf) * H + j) * 4 + q4] = wt2[(size_t)(8 * g + 2 * q4 + hf) * H + j];
    }
    const size_t o_wmu = put(d->W_mu, H), o_bmu = put(d->b_mu, 1), o_ls = put(d->log_sigma, 1);
    const size_t o_wbr = put(d->W_branch, (size_t)d->n_branch * H), o_bbr = put(d->b_branch, d->n_branch);
    hipError_t e;
    if ((e = hipMalloc(&pd.weights, host.size() * sizeof(float))) != hipSuccess ||
        (e = hipMemcpyAsync(pd.weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess) {
        err = std::string("hk_policy_attach: ") + hipGetErrorString(e);
        return HK_ERR_HIP;
    }
    PolicyParams& q = pd.q;
    q.in_dim = K0; q.obs_dim = obs_dim; q.stack = d->stack; q.hidden = H; q.n_layers = d->n_layers; q.n_branch = d->n_branch;
    q.normalize = d->normalize; q.deterministic = d->deterministic; q.seed = d->seed; q.index = index;
    {   // equal even chunks of at most PM_KC_MAX inputs
        const int nch = (K0 + PM_KC_MAX - 1) / PM_KC_MAX;
        q.kc = (K0 + nch - 1) / nch;
        q.kc = (q.kc + 7) & ~7;      // whole groups of 8 inputs per chunk (the float4 weight copy is read from group kb / 8 on)
        if (q.kc > PM_KC_MAX) q.kc = PM_KC_MAX;
        if (q.kc < H) q.kc = H;      // later layers run as one chunk of `hidden`
    }
    q.n_slots = n_slots;
    for (int j = 0; j < HK_MAX_AGENTS; j++) q.slots[j] = j < n_slots ? slots[j] : 0;
    q.mean = d->normalize ? pd.weights + o_mean : nullptr;
    q.std = d->normalize ? pd.weights + o_std : nullptr;
    for (int l = 0; l < HK_POLICY_MAX_LAYERS; l++) {
        q.Wt[l] = l < d->n_layers ? pd.weights + o_W[l] : nullptr;
        q.Wq[l] = l < d->n_layers ? reinterpret_cast<const float4*>(pd.weights + o_Wq[l]) : nullptr;
        q.b[l] = l < d->n_layers ? pd.weights + o_b[l] : nullptr;
    }
    q.W_mu = pd.weights + o_wmu; q.b_mu = pd.weights + o_bmu; q.log_sigma = pd.weights + o_ls;
    q.W_branch = pd.weights + o_wbr; q.b_branch = pd.weights + o_bbr;
    q.ring = nullptr; q.epoch = nullptr;
    if (E > 0) {
        const size_t pairs = (size_t)E * n_slots;
        if ((e = hipMalloc(&q.ring, pairs * K0 * sizeof(float))) != hipSuccess ||
            (e = hipMalloc(&q.epoch, pairs * sizeof(int32_t))) != hipSuccess ||
            (e = hipMemsetAsync(q.ring, 0, pairs * K0 * sizeof(float), stream)) != hipSuccess ||
            (e = hipMemsetAsync(q.epoch, 0xFF, pairs * sizeof(int32_t), stream)) != hipSuccess) {
            err = std::string("hk_policy_attach: ") + hipGetErrorString(e);
            return HK_ERR_HIP;
        }
    }
    pd.used = true;
    return HK_OK;
}

inline void policy_free(PolicyDevice& pd)
{
    if (pd.weights) (void)hipFree(pd.weights);
    if (pd.q.ring) (void)hipFree(pd.q.ring);
    if (pd.q.epoch) (void)hipFree(pd.q.epoch);
    pd = PolicyDevice{};
}

inline int policy_launch_mlp(const PolicyDevice& pd, int rows, const float* src, int w, unsigned long long decision,
                             int env_id_base, int A, float* mu_out, float* logit_out, float* act_steer, int* act_branch,
                             hipStream_t stream, std::string& err)
{
    if (rows <= 0) return HK_OK;
    const size_t lds = policy_lds_bytes(pd.q);
    // the kernels use ~68 KB of dynamic LDS, above the 64 KB default: raise the limit once per DEVICE (the attribute belongs to
    // the device's code object; a second handle on another device needs it too).  Guarded by a mutex: handles of different
    // devices may be driven from different host threads.
    {
        static std::mutex mu;
        static unsigned long long done_mask = 0ull;     // bit d: device d has the attribute
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !((done_mask >> dev) & 1ull)) {
            (void)hipFuncSetAttribute((const void*)policy_mlp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)policy_mlp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)policy_mlp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
        }
    }
    const dim3 grid((rows + PM_TILE - 1) / PM_TILE), block(PM_THREADS);
    if (pd.q.hidden <= 128)
        hipLaunchKernelGGL(policy_mlp_kernel<0>, grid, block, lds, stream, pd.q, rows, src, w, decision, env_id_base, A, mu_out, logit_out, act_steer, act_branch);
    else if (pd.q.hidden == 256)
        hipLaunchKernelGGL(policy_mlp_kernel<1>, grid, block, lds, stream, pd.q, rows, src, w, decision, env_id_base, A, mu_out, logit_out, act_steer, act_branch);
    else
        hipLaunchKernelGGL(policy_mlp_kernel<2>, grid, block, lds, stream, pd.q, rows, src, w, decision, env_id_base, A, mu_out, logit_out, act_steer, act_branch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("policy_mlp_kernel: ") + hipGetErrorString(e); return HK_ERR_HIP; }
    return HK_OK;
}

}  // namespace hk
