// hk_env_launch.h — host-side launches of the env kernels of ONE lane-group width (namespace hk::HK_GA_NS); hk_env_kernels.h
// picks the width from hk_config.num_agents and forwards here.
// (included once per lane-group width by hk_env_ga.h: no include guard)

namespace hk { namespace HK_GA_NS {

inline size_t mcts_req_bytes() { return sizeof(MctsReq); }
inline int mcts_searches_per_wave() { return 64; }
inline size_t mcts_lds_bytes(int ntab, int L, int na, int tier, int waves) { return mcts_search_lds_bytes(ntab, L, na, tier, waves); }
inline int mcts_root_words() { return MC_ROOT_WORDS; }
inline size_t game_doubles_per_ego() { return (size_t)GA * GP_FIELDS; }      // GameSoA: GA players x GP_FIELDS doubles
inline size_t queue_ints_per_set(size_t na) { return (size_t)(GA - 1) * na; }   // one queue per player count 2 .. GA

inline int launch_mcts_table(EnvDevice& d, int ego0, int ntab, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(mcts_table_kernel, dim3((ntab + 255) / 256), dim3(256), 0, stream, d.P, d.mcts, ego0);
    int rc = launch_check(err, "mcts_table_kernel");
    if (rc) return rc;
    const int rows = ntab / d.mcts.na;
    hipLaunchKernelGGL(mcts_order_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, d.P, d.mcts, ego0, rows);
    return launch_check(err, "mcts_order_kernel");
}

inline int launch_mcts_invalidate(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    if (!d.mcts.st) return HK_OK;
    const int n = cfg.num_envs * cfg.num_agents;
    hipLaunchKernelGGL(mcts_invalidate_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d.mcts, n);
    return launch_check(err, "mcts_invalidate_copy_kernel");
}

// Run the planner searches queued so far and hand the tick kernel the other (cleared) queue set.  A search requested on
// tick t must be finished before tick t + latency (> RUN_CAP, checked in env_create); the tick kernel advances an env by at
// most RUN_CAP ticks per round, so flushing every MCTS_FLUSH_ROUNDS rounds with (MCTS_FLUSH_ROUNDS + 1) * RUN_CAP < latency
// is early enough — and batching the requests of several rounds into one launch matters, because a launch takes as long
// as its slowest search however few searches it holds.
// (side != nullptr: the search kernels go to `side` — the caller has ordered it behind the tick launches that posted the requests — while the clearing
// of the queue set the tick kernel fills next stays on `stream`, ahead of those launches)
inline int flush_mcts_on(EnvDevice& d, hipStream_t stream, hipStream_t side, std::string& err)
{
    if (!d.mcts.st) return HK_OK;
    hipStream_t kstream = side ? side : stream;
    for (int c = 0; c < d.n_mcls; c++) {          // one launch per gameParams class: its tables ride in the workgroups' LDS, its agents' entries are searched
        const EnvDevice::MctsClass& K = d.mcls[c];
        MctsDev M = d.mcts;
        M.dt_tab = K.dt_tab; M.load_tab = K.load_tab; M.rad_tab = K.rad_tab; M.mask_tab = K.mask_tab; M.order_tab = K.order_tab; M.nv = K.nv; M.na = K.na; M.lds_tier = K.lds_tier; M.ntab = K.ntab;
        // one workgroup per CU, 4 waves (one per SIMD) while the grid is at most 1 024 waves, 8 beyond; the move tables ride in its LDS
        const int total_waves = (d.mcts.grid_lanes + 63) / 64;
        int waves = total_waves <= 1024 ? 4 : 8;
        // beside tick launches (side != nullptr): two waves per SIMD on HALF the CUs — a search workgroup's tables take ~100 KB of a CU's LDS, and
        // beside one a CU holds a single tick / B1 block instead of three; packed two to a SIMD the searches leave the other CUs to the ticks
        if (side && d.mcts_side_waves) waves = d.mcts_side_waves;
        while (waves > 4 && mcts_search_lds_bytes(K.ntab, d.P.L, K.na, K.lds_tier, waves) > 160 * 1024) waves -= 2;      // (8 karts: the path arrays of 8 waves do not fit beside the tables)
        const size_t lds = mcts_search_lds_bytes(K.ntab, d.P.L, K.na, K.lds_tier, waves);
        if (!d.mcts.lds_attr_set) {
            (void)hipFuncSetAttribute((const void*)mcts_search_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)mcts_search_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            d.mcts.lds_attr_set = 1;
        }
        if (K.lds_tier == 7) hipLaunchKernelGGL(mcts_search_kernel<true>, dim3((total_waves + waves - 1) / waves), dim3(waves * 64), lds, kstream, d.P, M, d.mset, K.ntab, K.agents);
        else hipLaunchKernelGGL(mcts_search_kernel<false>, dim3((total_waves + waves - 1) / waves), dim3(waves * 64), lds, kstream, d.P, M, d.mset, K.ntab, K.agents);
    }
    int rc = launch_check(err, "mcts_search_kernel");
    if (rc) return rc;
    d.mset ^= 1;
    d.mcts_rounds = 0;
    d.mcts_ticks = 0;
    if (hipMemsetAsync(d.mcts.qcnt + d.mset * 2, 0, 2 * sizeof(int), stream) != hipSuccess) { err = "mcts queue memset"; return HK_ERR_HIP; }
    return HK_OK;
}
inline int flush_mcts(EnvDevice& d, hipStream_t stream, std::string& err) { return flush_mcts_on(d, stream, nullptr, err); }

inline int launch_reset(EnvDevice& d, const int* dids, int cnt, int experiment_num, hipStream_t stream, std::string& err)
{
    const int threads = cnt * GA;
    hipLaunchKernelGGL(env_reset_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, d.P, d.agents, d.hot, d.slot_of, d.envs, dids, cnt, experiment_num,
                       d.mcts, d.mset, d.rw, d.status);
    int rc = launch_check(err, "env_reset_kernel");
    if (rc) return rc;
    return flush_mcts(d, stream, err);          // the first plans (T = 1.5 s in the reference)
}

// re-assign the lane groups by solve phase (see env_regroup_count_kernel); only meaningful with the 4-tick cadence
inline int launch_regroup(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    if (cfg.num_agents <= 2 || !d.perm) return HK_OK;
    const int E = cfg.num_envs;
    if (hipMemsetAsync(d.perm_counts, 0, 32 * sizeof(int), stream) != hipSuccess) { err = "regroup memset"; return HK_ERR_HIP; }
    hipLaunchKernelGGL(env_regroup_count_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, d.envs, E, d.perm_counts);
    hipLaunchKernelGGL(env_regroup_scatter_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, d.envs, d.envs_alt, d.hot, d.hot_alt, d.perm, d.perm_alt, d.slot_of,
                       E, d.perm_counts, inwave_now(d) ? 1 : 0);          // (in-wave solves: the envs that hold games spread over the waves, not packed)
    const int rc = launch_check(err, "env_regroup kernels");
    if (rc) return rc;           // (a launch that failed wrote nothing: the host pointers stay on the buffers that hold the state)
    // the move is physical: from here on the stream's kernels use the new buffers (everything is issued in stream order)
    std::swap(d.envs, d.envs_alt); std::swap(d.hot, d.hot_alt); std::swap(d.perm, d.perm_alt);
    d.regroup_mode = inwave_now(d) ? 1 : 0;
    return HK_OK;
}

// one round, part 1: the fused tick kernel (fills queue set round & 1)
inline int launch_run(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    const int s0 = d.slot1 > d.slot0 ? d.slot0 : 0, s1 = d.slot1 > d.slot0 ? d.slot1 : cfg.num_envs;
    const long long threads = (long long)(s1 - s0) * GA;
    const int arm = d.arm_ticks, guard = (d.guard_rounds_left > 0 && --d.guard_rounds_left == 0) ? 1 : 0;
    d.arm_ticks = 0;
    d.inwave = false;            // (launch_b1 decides for the round)
#define HK_RUN_T(MC, RWF, TRN, TL)                                                                                            \
    hipLaunchKernelGGL((env_run_kernel<MC, RWF, TRN, TL>), dim3((unsigned)((threads + 255) / 256)), dim3(256), TL ? d.tab_lds : 0, stream, d.P, d.agents, d.hot, d.envs,   \
                       d.results, GameSoA{d.games, (size_t)cfg.num_envs * cfg.num_agents}, d.queue_cnt, d.queue, d.round, d.act_steer, d.act_branch, d.lq_debug, d.status,     \
                       d.mcts, d.mset, d.rw, d.perm, d.game_stats, s0, s1, d.qbase, arm, guard)
#define HK_RUN(MC, RWF, TRN) do { if (d.tab_lds) HK_RUN_T(MC, RWF, TRN, true); else HK_RUN_T(MC, RWF, TRN, false); } while (0)
    bool train = d.P.training_reset != 0;
    for (int i = 0; i < cfg.num_agents; i++) train = train || d.P.training_agent[i] != 0;
#if HK_GA == 4
    if (d.fission) {
        // FISSION (hk_env_run.h): the tick kernel without phase B1; launch_b1 follows on the same stream
        const GameSoA G{d.games, (size_t)cfg.num_envs * cfg.num_agents};
        const unsigned blocks = (unsigned)((threads + 255) / 256);
#define HK_FIS_RUN(MC, TL) hipLaunchKernelGGL((env_run_kernel<MC, false, false, TL, true>), dim3(blocks), dim3(256), TL ? d.tab_lds : 0, stream, d.P, d.agents, d.hot, d.envs, \
                               d.results, G, d.queue_cnt, d.queue, d.round, d.act_steer, d.act_branch, d.lq_debug, d.status, d.mcts, d.mset, d.rw, d.perm,        \
                               d.game_stats, s0, s1, d.qbase, arm, guard)
        // (reward shaping / Training mode: phases A and C carry them, so the tick kernel has <.., HAS_RW, HAS_TRAIN, ..> twins; phase B1 knows neither)
#define HK_FIS_RUN3(MC, RWF, TRN, TL) hipLaunchKernelGGL((env_run_kernel<MC, RWF, TRN, TL, true>), dim3(blocks), dim3(256), TL ? d.tab_lds : 0, stream, d.P, d.agents, d.hot, d.envs, \
                               d.results, G, d.queue_cnt, d.queue, d.round, d.act_steer, d.act_branch, d.lq_debug, d.status, d.mcts, d.mset, d.rw, d.perm,        \
                               d.game_stats, s0, s1, d.qbase, arm, guard)
        if (train) { if (d.tab_lds) HK_FIS_RUN3(true, true, true, true); else HK_FIS_RUN3(true, true, true, false); }
        else if (d.rw.sec_time) {
            if (d.mcts.st) { if (d.tab_lds) HK_FIS_RUN3(true, true, false, true); else HK_FIS_RUN3(true, true, false, false); }
            else { if (d.tab_lds) HK_FIS_RUN3(false, true, false, true); else HK_FIS_RUN3(false, true, false, false); }
        }
        else if (d.mcts.st) { if (d.tab_lds) HK_FIS_RUN(true, true); else HK_FIS_RUN(true, false); }
        else { if (d.tab_lds) HK_FIS_RUN(false, true); else HK_FIS_RUN(false, false); }
#undef HK_FIS_RUN
#undef HK_FIS_RUN3
        d.b1_due = d.P.any_lqr != 0;
    } else
#endif
    if (train) HK_RUN(true, true, true);
    else if (d.mcts.st) { if (d.rw.sec_time) HK_RUN(true, true, false); else HK_RUN(true, false, false); }
    else { if (d.rw.sec_time) HK_RUN(false, true, false); else HK_RUN(false, false, false); }
#undef HK_RUN_T
#undef HK_RUN
    return launch_check(err, "env_run_kernel");
}

// one round, part 1b (FISSION): phase B1 of every env the tick launch parked at its solve tick
inline int launch_b1(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
#if HK_GA == 4
    if (!d.b1_due) return HK_OK;
    d.b1_due = false;
    // in-wave solves (hk_lq_spread.h lqs_inwave) once the field has spread — while it stands close (BULK_TICKS after a reset of every env) nearly every ego
    // holds a game and the queues + the pair solver's 32 games per wave are several times cheaper per game
    const int meter_fresh = ((d.meter_fresh >> (d.qbase >> 1)) & 1u) ? 2 : 0;
    d.meter_fresh &= ~(1u << (d.qbase >> 1));
    d.inwave = inwave_now(d);      // (HK_INWAVE=1, tests: in every round)
    const int s0 = d.slot1 > d.slot0 ? d.slot0 : 0, s1 = d.slot1 > d.slot0 ? d.slot1 : cfg.num_envs;
    const long long threads = (long long)(s1 - s0) * GA;
    const GameSoA G{d.games, (size_t)cfg.num_envs * cfg.num_agents};
    const unsigned blocks = (unsigned)((threads + 255) / 256);
#define HK_FIS_B1(TL, MC) hipLaunchKernelGGL((env_b1_kernel<TL, MC>), dim3(blocks), dim3(256), TL ? d.P.o_tmask : 0, stream, d.P, d.agents, d.hot, d.envs, G, d.queue_cnt, d.queue, \
                           d.round, d.lq_debug, d.status, d.mcts, d.perm, d.game_stats, s0, s1, d.qbase, d.mset, (d.inwave ? 1 : 0) | meter_fresh)
    // (b1_small: beside a search launch whose 4-wave workgroups hold 108.8 KB of EVERY CU's LDS, a B1 block with its 44.5 KB copy of the Complex-track tables
    // — 67 KB with the KartS staging — does not fit; the instantiation that reads the tables through L1 / L2 needs the 22.8 KB of staging only)
    if (d.mcts.st) { if (d.tab_lds && !d.b1_small) HK_FIS_B1(true, true); else HK_FIS_B1(false, true); }
    else { if (d.tab_lds) HK_FIS_B1(true, false); else HK_FIS_B1(false, false); }
#undef HK_FIS_B1
    return launch_check(err, "env_b1_kernel");
#else
    (void)d; (void)cfg; (void)stream; (void)err;
    return HK_OK;
#endif
}

// one round, part 2: the Riccati solves of the queued multi-player games, binned by player count
inline int launch_lqn(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    const int ngames = cfg.num_envs * cfg.num_agents;
    const int set = d.qbase + (d.round & 1);
    if (d.mcts.st && !d.mcts_defer && ++d.mcts_rounds >= MCTS_FLUSH_ROUNDS) { int rcm = flush_mcts(d, stream, err); if (rcm) return rcm; }
    const int* qc = d.queue_cnt + set * 16;
    const int* qu = d.queue + (size_t)set * queue_ints_per_set((size_t)ngames);
    int rc = HK_OK;
    d.lqn_launched = false;
    if (d.inwave) {
        // IN-WAVE (round 6): the B1 launch of this round solved its multi-player games itself (hk_env_run.h, hk_lq_spread.h lqs_inwave): nothing is queued
        if (d.last_solve_skippable && d.guard_rounds_left == 0) d.last_solve_skippable = false;
        d.round += 1;
        if (d.qbase == 0) d.call_ticks_issued = std::min(d.call_ticks_issued + (cfg.num_agents > 2 ? 4 : 1), d.call_ticks);
        return HK_OK;
    }
    d.lqn_launched = true;
    if (d.last_solve_skippable && d.guard_rounds_left == 0) {
        // The tick launch before this one was the call's last (its completion guard): an env that queued a game there would be parked, i.e.
        // the call incomplete, which the round count rules out.  The skip is verified ON THE DEVICE by that very launch: as the guard it
        // raises status bit 2 for any env it leaves with ticks or a phase (a parked env has phase 1 or 2), and the next getter reports it.  Nothing to solve: a one-tick call is 3 launches, not 4.
        d.last_solve_skippable = false;
        d.round += 1;
        if (d.qbase == 0) d.call_ticks_issued = std::min(d.call_ticks_issued + (cfg.num_agents > 2 ? 4 : 1), d.call_ticks);
        return HK_OK;
    }
    // Bulk or sparse?  The host cannot see the queues without a sync, but it knows how long ago the field stood on the start grid:
    // for BULK_TICKS after a reset of every env most egos hold a 2-player game (their row mate), later almost none does.
    const bool bulk = cfg.num_agents == 2 || d.ticks_since_reset + d.call_ticks_issued < BULK_TICKS;
    {
        // One launch for every game of up to 4 players (lqn_round_kernel): the 2-player queue goes to pairs of lanes, 32 games a wave,
        // no barriers (hk_lq2_pair.h); the 3- and 4-player queues to the matrix-core solver (hk_lq_mfma.h), one game per wave,
        // 1 024 workgroups per game size (one wave per SIMD) walking their queue grid-stride.
        const int sizes = cfg.num_agents > 2 ? std::min(cfg.num_agents, 4) - 2 : 0;     // player counts 3 .. min(A, 4)
        // (a spread field queues a few dozen games per launch: d.lqn_sparse_blocks workgroups per queue then — every wave of this kernel needs a SIMD's
        // whole register file, and in the two-stream schedule each one waits for a SIMD the other half's tick / B1 waves have left)
        const int per = bulk ? 1024 : d.lqn_sparse_blocks;
        const int n34 = sizes ? std::min(ngames, per) * sizes : 0;          // `per` waves per game size walk their queue grid-stride; the queue length picks the solver (lqn_round_kernel)
        const int n2 = std::min((ngames + 31) / 32, bulk ? 4096 : per);
        if (d.lqn_spread && !d.dense && (!bulk || d.b1_small)) {
            // A spread field (round 6): every queue on the lane-per-(player, row) solver (hk_lq_spread.h) — a third of the pair solver's registers, 2 - 11 KB of
            // LDS per wave, ~4 x shorter per lane: its waves start beside the other half's resident tick / B1 waves (and beside a planner's search waves)
            // instead of waiting for a CU to drain.  The pair / matrix-core kernel below keeps the rounds in which (nearly) every ego holds a game.
            const int s2 = std::min((ngames + 3) / 4, per), s34 = sizes ? std::min(ngames, std::max(per / 8, 16)) : 0;
            hipLaunchKernelGGL(lqn_spread_kernel, dim3(s2 + s34 * sizes), dim3(64), 0, stream, d.P, HotRef{d.hot, d.slot_of}, GameSoA{d.games, (size_t)ngames}, qc, qu, d.lq_debug, d.status,
                               s2, sizes >= 1 ? s34 : 0, sizes >= 2 ? s34 : 0, d.game_stats);
        } else
        if (d.b1_small)     // beside a search launch (hk_lq2_pair.h: SMALL)
            hipLaunchKernelGGL(lqn_round_small_kernel, dim3(n34 + n2), dim3(64), lqn_round_small_lds(), stream, d.P, HotRef{d.hot, d.slot_of}, GameSoA{d.games, (size_t)ngames}, qc, qu, d.lq_debug, d.status,
                               n34, sizes ? sizes : 1, n2, d.game_stats, LQN_BULK_GAMES);
        else
        hipLaunchKernelGGL(lqn_round_kernel, dim3(n34 + n2), dim3(64), 0, stream, d.P, HotRef{d.hot, d.slot_of}, GameSoA{d.games, (size_t)ngames}, qc, qu, d.lq_debug, d.status,
                           n34, sizes ? sizes : 1, n2, d.game_stats, LQN_BULK_GAMES);
        if ((rc = launch_check(err, "lqn_round_kernel"))) return rc;
    }
    if (cfg.num_agents > 4) {
#if HK_GA > 4
        {
            const int nbb = std::min((ngames + 1) / 2, 512);
            hipLaunchKernelGGL(lqn_big_kernel<5>, dim3(nbb * (cfg.num_agents > 5 ? 2 : 1)), dim3(64), 0, stream, d.P, HotRef{d.hot, d.slot_of}, GameSoA{d.games, (size_t)ngames}, qc, qu,
                               d.lq_debug, d.status, nbb, d.game_stats);
            if ((rc = launch_check(err, "lqn_big_kernel<5>"))) return rc;
            if (cfg.num_agents > 6) {
                hipLaunchKernelGGL(lqn_big_kernel<7>, dim3(nbb * (cfg.num_agents > 7 ? 2 : 1)), dim3(64), 0, stream, d.P, HotRef{d.hot, d.slot_of}, GameSoA{d.games, (size_t)ngames}, qc, qu,
                                   d.lq_debug, d.status, nbb, d.game_stats);
                if ((rc = launch_check(err, "lqn_big_kernel<7>"))) return rc;
            }
        }
#endif
    }
    d.round += 1;
    if (d.qbase == 0) d.call_ticks_issued = std::min(d.call_ticks_issued + (cfg.num_agents > 2 ? 4 : 1), d.call_ticks);      // a round retires at least one solve cadence
    return HK_OK;
}

inline int launch_observe(EnvDevice& d, const hk_config& cfg, uint32_t agent_mask, hipStream_t stream, std::string& err)
{
    const long long threads = (long long)cfg.num_envs * cfg.num_agents * OBS_LANES;
    // a block is only 16 agents: copying the tables into LDS pays for the Oval's 20 KB (+3 % on the RL workload), not for the
    // Complex track's 40 KB (the 8-agent workload lost 10 %), which keeps reading them through L1 / L2
#ifndef HK_OBS_LDS_MAX
#define HK_OBS_LDS_MAX (32 * 1024)     /* round 5: the Oval tables are 29 KB now; in LDS 71.2 -> 72.8 M on the RL workload */
#endif
    const int lds = (d.tab_lds && d.tab_lds <= HK_OBS_LDS_MAX) ? d.tab_lds : 0;
    const uint32_t mask = d.rw.hit_code ? 0xFFFFFFFFu : agent_mask;                  // the reward replay needs every agent's hit codes
#ifndef HK_OBS_BIG_BLOCK
#define HK_OBS_BIG_BLOCK 1024
#endif
#ifndef HK_OBS_LDS_BLOCK
#define HK_OBS_LDS_BLOCK 512      /* one copy of the staged tables per 32 agents: RL workload 75.5 -> 77.5 M (256: a copy per 16; 1 024: 76.0) */
#endif
    if (lds) hipLaunchKernelGGL((env_observe_kernel<true, HK_OBS_LDS_BLOCK>), dim3((unsigned)((threads + HK_OBS_LDS_BLOCK - 1) / HK_OBS_LDS_BLOCK)), dim3(HK_OBS_LDS_BLOCK), lds, stream, d.P, d.agents, d.hot, d.slot_of,
                                d.obs, d.rw.hit_code, mask);
    else if (d.tab_lds && d.tab_lds <= 60 * 1024)      // a long track: one copy of the tables per 64 agents
        hipLaunchKernelGGL((env_observe_kernel<true, HK_OBS_BIG_BLOCK>), dim3((unsigned)((threads + HK_OBS_BIG_BLOCK - 1) / HK_OBS_BIG_BLOCK)), dim3(HK_OBS_BIG_BLOCK), d.tab_lds, stream, d.P, d.agents, d.hot,
                           d.slot_of, d.obs, d.rw.hit_code, mask);
    else hipLaunchKernelGGL(env_observe_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d.P, d.agents, d.hot, d.slot_of, d.obs, d.rw.hit_code, mask);
    int rc = launch_check(err, "env_observe_kernel");
    if (rc || !d.rw.hit_code) return rc;
    // CollectObservations raised HitWall / HitOpponent events (HKA:580-598): replayed per env in agent / sensor order
    hipLaunchKernelGGL(reward_hits_kernel, dim3((cfg.num_envs + 127) / 128), dim3(128), 0, stream, d.P, d.agents, d.hot, d.slot_of, d.rw.hit_code);
    return launch_check(err, "reward_hits_kernel");
}

inline int launch_arm(EnvDevice& d, const hk_config& cfg, int n_ticks, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(env_arm_kernel, dim3((cfg.num_envs + 255) / 256), dim3(256), 0, stream, d.envs, cfg.num_envs, n_ticks, d.status);
    return launch_check(err, "env_arm_kernel");
}

inline int launch_done_check(EnvDevice& d, const hk_config& cfg, int lazy, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(env_check_kernel, dim3((cfg.num_envs + 255) / 256), dim3(256), 0, stream, d.envs, cfg.num_envs, d.status, lazy, d.P.guard_flag);
    return launch_check(err, "env_check_kernel");
}

inline int launch_rewards_read(EnvDevice& d, int cnt, float* reward, float* group_reward, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(rewards_read_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d.agents, cnt, reward, group_reward);
    return launch_check(err, "rewards_read_kernel");
}

// host-facing views of the device layout (hk_env_run.h: hot_gather_kernel ...)
inline int launch_hot_gather(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    const int n = cfg.num_envs * cfg.num_agents;
    hipLaunchKernelGGL(hot_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d.agents, d.hot, d.slot_of, n, cfg.num_agents);
    return launch_check(err, "hot_gather_kernel");
}
inline int launch_hot_scatter(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    const int n = cfg.num_envs * cfg.num_agents;
    hipLaunchKernelGGL(hot_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d.agents, d.hot, d.slot_of, n, cfg.num_agents);
    return launch_check(err, "hot_scatter_kernel");
}
inline int launch_envs_gather(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(envs_gather_kernel, dim3((cfg.num_envs + 255) / 256), dim3(256), 0, stream, d.envs_stage, d.envs, d.slot_of, cfg.num_envs);
    return launch_check(err, "envs_gather_kernel");
}
inline int launch_envs_scatter(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    hipLaunchKernelGGL(envs_scatter_kernel, dim3((cfg.num_envs + 255) / 256), dim3(256), 0, stream, d.envs_stage, d.envs, d.slot_of, cfg.num_envs);
    return launch_check(err, "envs_scatter_kernel");
}
inline size_t hot_tile_words(int E) { return hot_words(E, GA); }

inline const GaOps& make_ops()
{
    static const GaOps ops = {mcts_req_bytes, mcts_searches_per_wave, mcts_lds_bytes, mcts_root_words, game_doubles_per_ego, queue_ints_per_set,
                              launch_mcts_table, launch_mcts_invalidate, flush_mcts, flush_mcts_on, launch_reset, launch_regroup, launch_run, launch_b1, launch_lqn,
                              launch_observe, launch_arm, launch_done_check, launch_rewards_read, launch_hot_gather, launch_hot_scatter, launch_envs_gather,
                              launch_envs_scatter, hot_tile_words};
    return ops;
}

} }  // namespace hk::HK_GA_NS
