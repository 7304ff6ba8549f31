/*
 * hk_oracle_mcts.c — CPU ORACLE (test infrastructure): the MCTS high-level planner, restated from
 *   KM  = AI/MCTS/KartMCTS.cs            (tree search: constructSearchTree :50-77, getBestStatesSequence :108-123,
 *                                          UCTWeight :162-165, upperConfidenceStrategy :167-193, findLeaf :195-202,
 *                                          NextGaussian :221-240, simulate :242-283, backpropagate :285-293)
 *   KDG = AI/MCTS/KartDiscreteGame.cs    (DiscreteKartState.computeTOC :66-117, applyAction :122-167,
 *                                          DiscreteGameState.upNext :183-238, isOver :246-313, nextMoves :318-411,
 *                                          makeMove :416-443)
 *   HKA = AI/HierarchicalKartAgent.cs    (planWithMCTS :172-263, consumption in FixedUpdate :366-402)
 *   DPT = DiscretePositionTracker.cs     (radiusOfLane :153-158, distanceToTravel :163-175, tireLoad :180-192, Start :72-88)
 *   AK  = KartSystems/ArcadeKart.cs      (getMaxLateralGsForWear :517-520, getMaxSpeedForRadiusAndWear :536-547)
 * Structure follows the C#: every tree node owns a full copy of the discrete game state.  (The HIP kernel keeps ONE
 * running state per search and compact nodes instead; the two must still agree draw for draw.)
 *
 * Reference quirks kept on purpose: `player = count` with count never incremented (HKA:235, every kart state has
 * player 0); the velocity bucket loop breaks on its first pass (HKA:211-219: min 0, max = bucket); applyAction reads
 * newState.tireAge (still 0) for the TOC wear (KDG:150); integer division inside Mathf.Log (KM:164); foreach (int score
 * ...) truncation and score accumulators that are not reset per player (KDG:273-307); the missing `else` that doubles
 * score entries (KDG:255-261); the disabled collision filter (KDG:398).
 *
 * PARITY STATUS — "parity unpinned", by construction: the reference searches on a background thread until a wall-clock
 * budget expires and draws from System.Random / MathNet Normal.  Here: a fixed iteration budget, results visible a fixed
 * number of ticks after the request, Philox-4x32 draws.  List<T>.Sort is restated as the .NET introsort small-partition
 * rules (2: one compare-swap, 3: three compare-swaps lo/hi-1, lo/hi, hi-1/hi, 4..16: insertion sort).
 * Root reuse (HKA:66-67,175,265-283,660-669) is restated with REAL persistence here: the tree of an agent's last plan stays
 * allocated (hko_env.trees) and a replan that finds it alive searches it again.  (The HIP side keeps no trees: it rebuilds one by
 * replaying the searches it received, draw for draw — the two must agree.)  One bound is ours: a tree receives at most
 * HK_MCTS_MAX_ROOT_PHASES = 3 searches (the reference's count; a finishing background thread that re-installs a root after the main
 * thread reset CyclesRootProcessed could add more there); beyond that a replan starts a new tree.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hk_oracle.h"
#include "../include/hk_detmath.h"
#include "hk_oracle_internal.h"

#define MAXP HK_MAX_AGENTS
#define MAXA HK_MCTS_MAX_ACTIONS

typedef struct { int min_velocity, max_velocity, lane; } action_t;      /* DiscreteKartAction KDG:14-19 */

typedef struct {                                                         /* DiscreteKartState KDG:21-33 */
    int agent;            /* stands for `name` (unique per kart) */
    int team, section, timeAtSection, min_velocity, max_velocity, lane, tireAge, laneChanges, infeasible;
} kart_t;

typedef struct {                                                         /* DiscreteGameState KDG:170-178 */
    int P;
    kart_t k[MAXP];
    int initialSection, lastCompletedSection, finalSection;
} game_t;

typedef struct node {                                                    /* KartMCTSNode KM:19-38 */
    game_t state;
    struct node* parent;
    int n_children;
    int child_action[MAXA];        /* insertion order: index into the canonical action list */
    struct node* child[MAXA];
    float totalValue;
    int numEpisodes;
} node_t;

typedef struct {
    const hko_env* e;
    int ego;                        /* the planning agent: gameParams come from it */
    int bucket, precision;
    /* draws */
    uint32_t key0, key1, c1, c2;
    uint32_t draw;
    /* node arena of the search in progress (a chunk of the agent's tree) */
    node_t* pool; int pool_n, pool_cap;
} search_t;

/* currentRoot (HKA:66): the tree of an agent's last plan, kept while root_live.  Every search adds one chunk of nodes. */
struct hko_tree {
    node_t* chunk[HK_MCTS_MAX_ROOT_PHASES];
    int n_chunks;
    int P;                          /* players of the root game */
    uint8_t player_agent[MAXP];
};

/* ---- draws (stand-ins for System.Random / MathNet.Numerics Normal; see header) */
static void draw4(search_t* S, uint32_t r[4]) { philox4x32(S->draw++, S->c1, S->c2, 0x4D435453u, S->key0, S->key1, r); }
static int rand_next(search_t* S, int n)
{   /* random.Next(n) */
    uint32_t r[4]; draw4(S, r);
    return (int)(((uint64_t)r[0] * (uint64_t)n) >> 32);
}
static float normal_sample(search_t* S)
{   /* normalDist.Sample(): Box-Muller on two uniforms */
    uint32_t r[4]; draw4(S, r);
    float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f);
    float u2 = u01(r[1]);
    return sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);
}
static float next_gaussian_bounded(search_t* S, float mean, float sd, float lo, float hi)
{   /* KM:225-240 */
    float x; int attempts = 0;
    do { x = mean + normal_sample(S) * sd; attempts += 1; } while ((x < lo || x > hi) && attempts < 10);
    if (attempts == 10 && (x < lo || x > hi)) return mean;
    return x;
}
static int round_to_int(float v)
{   /* Mathf.RoundToInt = (int)Math.Round(v): half to even */
    return (int)__builtin_rintf(v);
}

/* ---- track / kart helpers */
static const hk_section* sec_of(const hko_env* e, int section) { return &e->sec[section % e->L]; }
static int straight(const hko_env* e, int section) { return sec_of(e, section)->track_inside_radius == 0.0f; }
static float lane_radius(const hk_section* s, int lane)
{   /* DPT.Start :72-88 radiuses[lane - 1] */
    int q = s->left_turn ? (lane - 1) : (4 - lane);
    return s->track_inside_radius + s->track_width * ((float)q / 4.0f);
}
static float radius_of_lane(const hko_env* e, int section, int l0, int l1)
{   /* DPT:153-158 via REC.computeAvgSectionRadius */
    const hk_section* s = sec_of(e, section);
    if (s->track_inside_radius == 0.0f) return 0.0f;
    return (lane_radius(s, l0) + lane_radius(s, l1)) / 2.0f;
}
static float distance_to_travel(const hko_env* e, int section, int l0, int l1)
{   /* DPT:163-175 */
    const hk_section* s = sec_of(e, section);
    if (s->track_inside_radius == 0.0f) {
        float w = ((float)abs(l0 - l1) * 1.0f / 3.0f) * s->track_width;
        return sqrtf(w * w + s->track_length * s->track_length);
    }
    float avg = radius_of_lane(e, section, l0, l1);
    return (HK_PI_F / 180.0f) * s->turn_degrees * avg;
}
static float tire_load(const hko_env* e, int section, float velocity, int l0, int l1)
{   /* DPT:180-192 */
    if (straight(e, section)) return distance_to_travel(e, section, l0, l1) * 0.01f;
    float gs = (velocity * velocity) / radius_of_lane(e, section, l0, l1);
    return gs * distance_to_travel(e, section, l0, l1) * 0.01f;
}
static float max_speed_radius_wear(const hko_env* e, float radius, float wear)
{   /* AK:536-547 */
    const hk_kart_stats* st = &e->cfg.stats;
    if (radius == 0.0f) return st->TopSpeed;
    float gsw = (1 - wear) * (st->MaxGs - st->MinGs) + st->MinGs;                     /* AK:517-520 */
    float v = sqrtf(gsw * 9.81f * fabsf(radius));
    if (isinf(v) || isnan(v)) v = st->TopSpeed;
    return v < 0.0001f ? 0.0001f : (v > st->TopSpeed ? st->TopSpeed : v);
}
static float avg_velocity(const kart_t* k) { return (1.0f * (float)(k->min_velocity + k->max_velocity)) / 2.0f; }   /* KDG:58-61 */

static float compute_toc(const hko_env* e, float distance, float radius, float tireWear, float initV, float finalV)
{   /* KDG:66-117 */
    const float acc = e->cfg.stats.Acceleration, brk = e->cfg.stats.Braking;
    if (finalV > initV && (finalV * finalV - initV * initV) / (2 * acc) > distance) return -1.0f;
    if (initV > finalV && (initV * initV - finalV * finalV) / (2 * brk) > distance) return -1.0f;
    float vmax = max_speed_radius_wear(e, radius, tireWear);
    float t1 = vmax >= initV ? (vmax - initV) / acc : (initV - vmax) / brk;
    float t3 = vmax >= finalV ? (vmax - finalV) / brk : (finalV - vmax) / acc;
    float x1 = 0.5f * (initV + vmax) * t1;
    float x3 = 0.5f * (finalV + vmax) * t3;
    float x2 = distance - x1 - x3;
    float t2 = x2 / vmax;
    if ((double)t2 > 0.001) return t1 + t2 + t3;
    else if (initV <= vmax) {
        float ms = sqrtf((2 * distance * -brk * acc + -brk * initV * initV - acc * finalV * finalV) / (-acc - brk));
        t1 = (ms - initV) / acc;
        t3 = (ms - finalV) / brk;
        return t1 + t3;
    }
    return -1.0f;
}

static kart_t apply_action(const search_t* S, const kart_t* k, const action_t* a)
{   /* KDG:122-167 */
    const hko_env* e = S->e;
    kart_t n; memset(&n, 0, sizeof(n));
    n.agent = k->agent; n.team = k->team;
    n.section = k->section + 1;
    n.min_velocity = a->min_velocity; n.max_velocity = a->max_velocity; n.lane = a->lane;
    if (straight(e, k->section) != straight(e, k->section + 1)) n.laneChanges = 0;
    else if (n.lane != k->lane) n.laneChanges = k->laneChanges + abs(n.lane - k->lane);
    else n.laneChanges = k->laneChanges;
    float dist = distance_to_travel(e, k->section, k->lane, a->lane);
    float rad = radius_of_lane(e, k->section, k->lane, a->lane);
    /* newState.tireAge is still 0 here (KDG:150) */
    int timeUpdate = (int)(compute_toc(e, dist, rad, (float)n.tireAge / 10000.0f, avg_velocity(k), avg_velocity(&n)) * (float)S->precision);
    if (timeUpdate < 0) n.infeasible = 1;
    n.timeAtSection = k->timeAtSection + timeUpdate;
    float load = tire_load(e, k->section, (float)a->max_velocity, k->lane, a->lane);
    n.tireAge = (int)(((float)k->tireAge / 10000.0f + load * e->cfg.stats.TireWearFactor) * 10000.0f);
    return n;
}

/* ---- DiscreteGameState */
static int cmp_kart(const kart_t* a, const kart_t* b)
{   /* the Comparison of KDG:186-223 */
    if (a->section < b->section) return -1;
    if (a->section > b->section) return 1;
    if (a->timeAtSection < b->timeAtSection) return -1;
    if (a->timeAtSection == b->timeAtSection) {
        float va = avg_velocity(a), vb = avg_velocity(b);
        if (va > vb) return -1;
        if (va == vb) return 0;
        return 1;
    }
    return 1;
}
static void swap_if_greater(int* o, const game_t* g, int i, int j)
{
    if (i != j && cmp_kart(&g->k[o[i]], &g->k[o[j]]) > 0) { int t = o[i]; o[i] = o[j]; o[j] = t; }
}
static int up_next(const game_t* g)
{   /* KDG:183-238; List.Sort = introsort, partitions <= 16 */
    int o[MAXP];
    for (int i = 0; i < g->P; i++) o[i] = i;
    if (g->P == 2) swap_if_greater(o, g, 0, 1);
    else if (g->P == 3) { swap_if_greater(o, g, 0, 1); swap_if_greater(o, g, 0, 2); swap_if_greater(o, g, 1, 2); }
    else if (g->P > 3) {
        for (int i = 0; i < g->P - 1; i++) {                       /* InsertionSort */
            int t = o[i + 1], j = i;
            while (j >= 0 && cmp_kart(&g->k[t], &g->k[o[j]]) < 0) { o[j + 1] = o[j]; j--; }
            o[j + 1] = t;
        }
    }
    for (int i = 0; i < g->P; i++)
        if (g->k[o[i]].section != g->lastCompletedSection + 1) return o[i];   /* .Equals finds the same kart: names are unique */
    return -1;
}

static int canonical_actions(const search_t* S, action_t* out)
{   /* KDG:325-338: velocity-major, lane-minor */
    int n = 0;
    const int vmax = (int)S->e->max_speed;
    for (int i = 6; i < vmax; i += S->bucket)
        for (int j = 1; j < 5; j++) {
            if (n >= MAXA) return n;
            out[n].min_velocity = i;
            out[n].max_velocity = (i + S->bucket) < vmax ? (i + S->bucket) : vmax;
            out[n].lane = j;
            n++;
        }
    return n;
}

/* KDG:318-411: indices (into the canonical list) of the legal actions of the player who is up next, in list order */
static int next_moves(const search_t* S, const game_t* g, int* legal)
{
    const hko_env* e = S->e;
    int np = up_next(g);
    const kart_t* cur = &g->k[np];
    action_t all[MAXA];
    int na = canonical_actions(S, all), n = 0;
    for (int i = 0; i < na; i++) {
        const action_t* a = &all[i];
        if (straight(e, cur->section) && cur->laneChanges + abs(a->lane - cur->lane) > e->cfg.max_lane_changes) continue;
        float radius = radius_of_lane(e, cur->section, cur->lane, a->lane);
        if (max_speed_radius_wear(e, radius, (float)cur->tireAge / 10000.0f) < (float)a->min_velocity) continue;
        kart_t ap = apply_action(S, cur, a);
        if (ap.infeasible) continue;
        legal[n++] = i;
    }
    return n;     /* the collision filter (KDG:385-409) cannot reject anything (`false &&`) */
}

static void make_move(const search_t* S, const game_t* g, const action_t* a, game_t* out)
{   /* KDG:416-443 */
    *out = *g;
    int np = up_next(g);
    out->k[np] = apply_action(S, &g->k[np], a);
    int allAhead = 1;
    for (int i = 0; i < out->P; i++) allAhead &= out->k[i].section > g->lastCompletedSection;
    if (allAhead) out->lastCompletedSection += 1;
}

/* KDG:246-313. returns 1 when over; scores[] as the reference's List<float> (its length can exceed P) */
static int is_over(const search_t* S, const game_t* g, float* scores)
{
    int legal[MAXA];
    if (next_moves(S, g, legal) == 0) {
        int noMove = up_next(g), n = 0;
        for (int i = 0; i < g->P; i++) {
            if (i == noMove || g->k[i].team == g->k[noMove].team) scores[n++] = 0.0f;
            scores[n++] = 0.5f;                                    /* no `else` */
        }
        return 1;
    }
    if (g->lastCompletedSection != g->finalSection) return 0;
    if (g->P > 1) {
        const float tsm = 0.75f;                                   /* RacingEnvController.TeamScoreRewardMultiplier (REC:88) */
        float maxScore = (float)S->precision * -1000.0f, minScore = (float)S->precision * 1000.0f;
        float raw[MAXP];
        float teamScore = 0.0f, opponentScore = 0.0f;
        int teamCount = 0, opponentCount = 0;                      /* not reset per player */
        for (int s = 0; s < g->P; s++) {
            for (int o = 0; o < g->P; o++) {
                if (s == o) teamScore += (float)g->k[o].timeAtSection;
                else if (g->k[s].team == g->k[o].team) { teamScore += (float)g->k[o].timeAtSection * tsm; teamCount += 1; }
                else { opponentScore += (float)g->k[o].timeAtSection; opponentCount += 1; }
            }
            float score = opponentScore * (((float)teamCount * tsm + 1.0f) / ((float)opponentCount * 1.0f)) - teamScore;
            raw[s] = score;
            /* Math.Max / Math.Min return NaN when either argument is NaN (no opponents nearby: 0 * inf) */
            maxScore = (isnan(maxScore) || isnan(score)) ? NAN : (maxScore > score ? maxScore : score);
            minScore = (isnan(minScore) || isnan(score)) ? NAN : (minScore < score ? minScore : score);
        }
        for (int s = 0; s < g->P; s++) {
            /* foreach (int score in scores): float -> int truncation; NaN / out of range convert to int.MinValue (x86) */
            int si = (raw[s] >= -2147483648.0f && raw[s] < 2147483648.0f) ? (int)raw[s] : (-2147483647 - 1);
            scores[s] = ((float)si - minScore) * 1.0f / (maxScore - minScore);
        }
        return 1;
    }
    scores[0] = (float)(S->e->cfg.max_episode_steps - g->k[0].timeAtSection / S->e->cfg.max_episode_steps);   /* int / int */
    return 1;
}

/* ---- the tree */
static node_t* new_node(search_t* S, const game_t* st, node_t* parent)
{
    if (S->pool_n >= S->pool_cap) return NULL;
    node_t* n = &S->pool[S->pool_n++];
    n->state = *st; n->parent = parent; n->n_children = 0; n->totalValue = 0.0f; n->numEpisodes = 0;
    return n;
}
static node_t* child_for(node_t* n, int action)
{
    for (int i = 0; i < n->n_children; i++) if (n->child_action[i] == action) return n->child[i];
    return NULL;
}
static float uct_weight(const node_t* n)
{   /* KM:162-165 */
    return (n->totalValue / (float)n->numEpisodes) + sqrtf(1.0f) * hk_logf((float)(n->parent->numEpisodes / n->numEpisodes));
}
static node_t* upper_confidence(search_t* S, node_t* n)
{   /* KM:167-193; Dictionary enumeration = insertion order */
    int index = rand_next(S, n->n_children);
    node_t* best = n->child[index];
    float best_uct = uct_weight(best);
    for (int i = 0; i < n->n_children; i++) {
        float u = uct_weight(n->child[i]);
        if (u > best_uct) { best_uct = u; best = n->child[i]; }
    }
    return best;
}
static node_t* find_leaf(search_t* S, node_t* root)
{   /* KM:195-202 */
    int legal[MAXA];
    while (root->n_children > 0 && root->n_children == next_moves(S, &root->state, legal)) root = upper_confidence(S, root);
    return root;
}

typedef struct { int idx, dt, maxv, dlane, slane; } okey_t;
static int okey_less(const okey_t* a, const okey_t* b)
{   /* OrderBy(dt).ThenByDescending(max_velocity).ThenBy(|dlane|).ThenBy(sign * lane), stable */
    if (a->dt != b->dt) return a->dt < b->dt;
    if (a->maxv != b->maxv) return a->maxv > b->maxv;
    if (a->dlane != b->dlane) return a->dlane < b->dlane;
    if (a->slane != b->slane) return a->slane < b->slane;
    return 0;
}

/* KM:242-283.  returns the terminal node (NULL if the arena ran out) and its scores */
static node_t* simulate(search_t* S, node_t* leaf, float* scores)
{
    action_t all[MAXA];
    canonical_actions(S, all);
    while (1) {
        if (is_over(S, &leaf->state, scores)) return leaf;
        const game_t* st = &leaf->state;
        int np = up_next(st);
        int sign;
        { int ol = sec_of(S->e, st->lastCompletedSection)->optimal_lane; sign = ol == 1 ? 1 : (ol == 4 ? -1 : 0); }   /* DPT:221-231 */
        int legal[MAXA];
        int n = next_moves(S, st, legal);
        okey_t key[MAXA];
        for (int i = 0; i < n; i++) {
            game_t tmp; make_move(S, st, &all[legal[i]], &tmp);
            key[i].idx = legal[i];
            key[i].dt = tmp.k[np].timeAtSection - st->k[np].timeAtSection;
            key[i].maxv = all[legal[i]].max_velocity;
            key[i].dlane = abs(all[legal[i]].lane - st->k[np].lane);
            key[i].slane = sign * all[legal[i]].lane;
        }
        for (int i = 1; i < n; i++) {                              /* stable insertion sort */
            okey_t t = key[i]; int j = i - 1;
            while (j >= 0 && okey_less(&t, &key[j])) { key[j + 1] = key[j]; j--; }
            key[j + 1] = t;
        }
        int index;
        if (n > 2) index = round_to_int(fabsf(next_gaussian_bounded(S, 0.0f, (float)n / 6.0f, -(float)n + 1.0f, (float)n - 1.0f)));
        else index = rand_next(S, n);
        int move = key[index].idx;
        node_t* c = child_for(leaf, move);
        if (!c) {
            game_t nx; make_move(S, st, &all[move], &nx);
            c = new_node(S, &nx, leaf);
            if (!c) return NULL;
            leaf->child_action[leaf->n_children] = move; leaf->child[leaf->n_children] = c; leaf->n_children++;
        }
        leaf = c;
    }
}

static void backpropagate(node_t* n, const float* result)
{   /* KM:285-293 */
    while (n) {
        n->totalValue += result[up_next(&n->state)];
        n->numEpisodes += 1;
        n = n->parent;
    }
}

size_t hko_mcts_tree_bytes(void) { return sizeof(struct hko_tree); }
static void tree_free(struct hko_tree* t)
{
    for (int c = 0; c < t->n_chunks; c++) free(t->chunk[c]);
    memset(t, 0, sizeof(*t));
}
void hko_mcts_trees_free(hko_env* e)
{
    if (!e->trees) return;
    for (size_t i = 0; i < (size_t)e->E * e->A; i++) tree_free(&e->trees[i]);
    free(e->trees); e->trees = NULL;
}
void hko_mcts_tree_drop(hko_env* e, int env, int ego) { if (e->trees) tree_free(&e->trees[(size_t)env * e->A + ego]); }

/* planWithMCTS HKA:172-284 + constructSearchTree KM:50-106 + getBestStatesSequence; result -> plan.
 * reuse = 0: the game is built from the karts' current state and searched in a new tree (HKA:175-263);
 * reuse = 1: the agent's existing root is searched again (HKA:265-283).
 * steer_seen: m_FinalStats.Steer of every kart as this plan sees it (NULL: the karts' current values; the reset passes stale ones). */
void hko_mcts_search(hko_env* e, int env, int ego, int iterations, int reuse, const float* steer_seen, hk_mcts_plan* plan)
{
    const hk_config* cfg = &e->cfg;
    const int A = e->A;
    const hk_agent_state* ags = &e->ag[(size_t)env * A];
    const hk_mcts_state* ms = &e->mcts[(size_t)env * A];
    struct hko_tree* T = &e->trees[(size_t)env * A + ego];
    search_t S; memset(&S, 0, sizeof(S));
    S.e = e; S.ego = ego;
    S.bucket = cfg->velocity_bucket_size[ego]; S.precision = cfg->time_precision[ego];
    S.key0 = cfg->mcts_seed; S.key1 = (uint32_t)(cfg->env_id_base + env) * (uint32_t)A + (uint32_t)ego;
    S.c1 = (uint32_t)e->es[env].episode_steps; S.c2 = (uint32_t)e->es[env].episodes_done;
    node_t* root;
    memset(plan, 0, sizeof(*plan));
    if (!reuse) {
        tree_free(T);
        game_t g; memset(&g, 0, sizeof(g));
        int nearby[MAXP], furthest = ego;
        int initialSection = ags[ego].section_index;
        for (int i = 0; i < A; i++) {                                                        /* :180-191 */
            if (abs(ags[i].section_index - ags[ego].section_index) < cfg->section_window[ego]) {
                nearby[g.P++] = i;
                if (ags[i].section_index > initialSection) initialSection = ags[i].section_index;
                if (initialSection == ags[i].section_index) furthest = i;
            }
        }
        for (int p = 0; p < g.P; p++) {
            const int ai = nearby[p];
            const hk_agent_state* a = &ags[ai];
            T->player_agent[p] = (uint8_t)ai;
            kart_t* k = &g.k[p];
            k->agent = ai; k->team = cfg->team_of[ai];
            k->min_velocity = 0;                                                             /* :211-219: the loop breaks at i = 0 */
            k->max_velocity = S.bucket < (int)e->max_speed ? S.bucket : (int)e->max_speed;
            k->section = initialSection;
            k->timeAtSection = 0;
            if (a->section_index != initialSection) {                                        /* :221-224 */
                const int ring = a->section_index & (HK_MCTS_SECTIME_RING - 1);
                k->timeAtSection = (int)((float)(ms[ai].sec_time[ring] - ms[furthest].sec_time[ring]) * cfg->dt * (float)S.precision);
            }
            k->lane = a->lane;
            k->tireAge = (int)((cfg->stats.MaxSteer - (steer_seen ? steer_seen[ai] : a->final_steer)) / (cfg->stats.MaxSteer - cfg->stats.MinSteer) * 10000.0f);   /* :236 */
            k->laneChanges = a->lane_changes;
            k->infeasible = 0;
        }
        g.initialSection = initialSection; g.lastCompletedSection = initialSection;
        g.finalSection = initialSection + cfg->tree_search_depth[ego];
        T->P = g.P;
        /* arena: every iteration adds at most (depth x players) nodes */
        S.pool_cap = 1 + iterations * (cfg->tree_search_depth[ego] * g.P + 1);
        S.pool = (node_t*)malloc(sizeof(node_t) * (size_t)S.pool_cap);
        T->chunk[T->n_chunks++] = S.pool;
        root = new_node(&S, &g, NULL);
    } else {
        root = &T->chunk[0][0];                                                              /* constructSearchTree(currentRoot, T: 0.9) KM:79-106 */
        S.pool_cap = iterations * (cfg->tree_search_depth[ego] * T->P + 1);
        S.pool = (node_t*)malloc(sizeof(node_t) * (size_t)S.pool_cap);
        T->chunk[T->n_chunks++] = S.pool;
    }
    plan->n_players = T->P;
    for (int p = 0; p < T->P; p++) plan->player_agent[p] = T->player_agent[p];
    float scores[2 * MAXP];
    for (int it = 0; it < iterations; it++) {                                            /* KM:50-77 with an iteration budget */
        node_t* leaf = find_leaf(&S, root);
        node_t* end = simulate(&S, leaf, scores);
        if (!end) break;
        backpropagate(end, scores);
    }
    /* getBestStatesSequence KM:108-123 */
    node_t* n = root;
    while (n->n_children > 0) {
        n = upper_confidence(&S, n);
        int all_at = 1;
        for (int p = 0; p < T->P; p++) all_at &= n->state.k[p].section == n->state.lastCompletedSection;
        if (all_at && plan->n_states < HK_MCTS_MAX_DEPTH) {
            int q = plan->n_states++;
            plan->section[q] = n->state.lastCompletedSection;
            for (int p = 0; p < T->P; p++) { plan->lane[q][p] = (uint8_t)n->state.k[p].lane; plan->vel[q][p] = (uint8_t)n->state.k[p].max_velocity; }
        }
    }
}

/* The replan of HKA.FixedUpdate :330-350 / initialPlan :84-96 for one agent: decide between a new tree, the old root again, or no
 * plan at all (CyclesRootProcessed >= 3), run the search and schedule its result.  (`t` is always finished here: a search lasts
 * less than the 100 ticks between two replans.) */
void hko_mcts_request(hko_env* e, int env, int ego, int iterations, int latency, const float* steer_seen)
{
    hk_mcts_state* m = &e->mcts[(size_t)env * e->A + ego];
    int kind;
    if (!m->root_live) kind = 1;                                                             /* HKA:175 */
    else if (m->root_cycles < 3) kind = m->root_phases < HK_MCTS_MAX_ROOT_PHASES ? 2 : 1;    /* HKA:265 (bound: see the header) */
    else return;
    hko_mcts_search(e, env, ego, iterations, kind == 2, steer_seen, &m->pend);
    m->root_phases = kind == 2 ? m->root_phases + 1 : 1;
    m->pend_kind = kind;
    m->searches += 1;
    m->ready_step = e->es[env].episode_steps + latency;
}

/* the search thread ends (HKA:250-253 / :271-273): bestStates, currentRoot and CyclesRootProcessed are written */
void hko_mcts_promote(hk_mcts_state* m)
{
    m->best = m->pend; m->ready_step = -1;
    if (m->pend_kind == 1) { m->root_live = 1; m->root_cycles = 1; }
    else if (m->pend_kind == 2) { m->root_live = 1; m->root_cycles += 1; }
    m->pend_kind = 0;
}

/* HKA.FixedUpdate :366-402: every tick, copy bestStates into the agent's own plan and its beliefs about the others */
void hko_mcts_consume(hko_env* e, int env, int i)
{
    const int A = e->A, L = e->L;
    hk_agent_state* a = &e->ag[(size_t)env * A + i];
    hk_mcts_state* m = &e->mcts[(size_t)env * A + i];
    const hk_mcts_plan* b = &m->best;
    for (int q = 0; q < b->n_states; q++)
        for (int p = 0; p < b->n_players; p++) {
            const int who = b->player_agent[p], sec = b->section[q];
            if (who == i) {
                if (sec > a->section_index + (a->section_index == 0 ? 0 : 1)) {
                    a->plan_lane[sec % L] = b->lane[q][p];
                    a->plan_vel[sec % L] = (float)b->vel[q][p];
                }
            } else {
                m->belief_lane[who][sec % L] = b->lane[q][p];
                m->belief_vel[who][sec % L] = b->vel[q][p];
            }
        }
}
