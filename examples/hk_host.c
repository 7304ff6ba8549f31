/* hk_host.c — a host that is neither Python nor ctypes: plain C against include/hk.h, linked to libhk.so the way a C#
 * [DllImport] shim binds it.  It reads a configuration blob (hk_config + its track arrays, written by
 * tests/test_c_host_gpu.py), runs a race batch through the C ABI and dumps the agent records, so the test can compare them
 * with the CPU oracle.  Usage: hk_host <config.bin> <ticks per call> <calls> <out.bin>
 * Build: gcc -O1 -I include -o hk_host examples/hk_host.c -L hierarchicalkarting_amd -lhk -Wl,-rpath,$PWD/hierarchicalkarting_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hk.h"

static void* read_exact(FILE* f, size_t bytes)
{
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "hk_host: short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s config.bin ticks calls out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    hk_config* cfg = (hk_config*)read_exact(f, sizeof(hk_config));
    if (cfg->abi_version != HK_ABI_VERSION) { fprintf(stderr, "hk_host: ABI %d != %d\n", cfg->abi_version, HK_ABI_VERSION); return 2; }
    cfg->sections = (const hk_section*)read_exact(f, sizeof(hk_section) * (size_t)cfg->num_sections);
    cfg->walls = (const hk_wall_seg*)read_exact(f, sizeof(hk_wall_seg) * (size_t)cfg->num_walls);
    fclose(f);
    const int ticks = atoi(argv[2]), calls = atoi(argv[3]);

    hk_handle h = NULL;
    int rc = hk_create(cfg, &h);
    if (rc != HK_OK) { fprintf(stderr, "hk_create: %d %s\n", rc, hk_last_error(NULL)); return rc == HK_ERR_NO_DEVICE ? 3 : 1; }
    if ((rc = hk_reset(h, NULL, 0, -1)) != HK_OK) { fprintf(stderr, "hk_reset: %d %s\n", rc, hk_last_error(h)); return 1; }

    const size_t na = (size_t)cfg->num_envs * (size_t)cfg->num_agents;
    hk_agent_state* st = (hk_agent_state*)malloc(na * sizeof(hk_agent_state));
    float* obs = (float*)malloc(na * (size_t)hk_obs_dim(h) * sizeof(float));
    FILE* out = fopen(argv[4], "wb");
    if (!st || !obs || !out) { fprintf(stderr, "hk_host: alloc / open failed\n"); return 2; }
    for (int c = 0; c < calls; c++) {
        if ((rc = hk_step(h, ticks)) != HK_OK) { fprintf(stderr, "hk_step: %d %s\n", rc, hk_last_error(h)); return 1; }
        if ((rc = hk_get_agent_state(h, st)) != HK_OK) { fprintf(stderr, "hk_get_agent_state: %d %s\n", rc, hk_last_error(h)); return 1; }
        if ((rc = hk_get_observations(h, obs)) != HK_OK) { fprintf(stderr, "hk_get_observations: %d %s\n", rc, hk_last_error(h)); return 1; }
        fwrite(st, sizeof(hk_agent_state), na, out);
        fwrite(obs, sizeof(float), na * (size_t)hk_obs_dim(h), out);
    }
    fclose(out);
    hk_episode_result* res = (hk_episode_result*)malloc(na * sizeof(hk_episode_result));
    if (hk_get_episode_results(h, res) != HK_OK) return 1;
    printf("hk_host: %d envs x %d agents, %d x %d ticks, obs_dim %d, first kart px %.6f pz %.6f section %d\n", cfg->num_envs,
           cfg->num_agents, calls, ticks, hk_obs_dim(h), st[0].px, st[0].pz, st[0].section_index);
    hk_destroy(h);
    return 0;
}
