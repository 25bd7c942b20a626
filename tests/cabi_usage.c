/* A plain C99 translation unit that uses both C ABIs the way a cgo shim would: it is compiled and linked by
 * tests/test_cabi.py::test_headers_are_plain_c, and run there without a GPU (every device entry point must then fail loudly with KS_ERR_DEVICE).
 * Nothing here is C++: plain pointers and sizes, no torch types. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ksolve.h"
#include "kshost.h"

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: cabi_usage <file.ksp> [file.envblock]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* text = (char*)malloc((size_t)n + 1); if (fread(text, 1, (size_t)n, f) != (size_t)n) return 2; text[n] = 0; fclose(f);

  void* env = NULL;                                     /* the environment: instance types, provisioners, nodes (PODS 0) */
  if (ksh_parse(text, (size_t)n, &env) != KS_OK) { fprintf(stderr, "parse: %s\n", ksh_last_error()); return 1; }

  /* one pod through the binary door: namespace "default", no labels / selectors / affinity / tolerations, one container asking for cpu=100m */
  const char strs[] = "default" "cpu" "pod-a";
  const uint32_t str_off[4] = {0, 7, 10, 15};
  const uint32_t words[] = {0 /*ns*/, 0 /*labels*/, 0 /*nodeSelector*/, 0 /*required terms*/, 0 /*preferred terms*/, 0 /*tolerations*/,
                            1 /*containers*/, 1 /*requests*/, 1 /*"cpu"*/, 100, 0 /*100 milli*/, 0 /*limits*/, 0 /*ports*/,
                            0 /*init containers*/, 0 /*spread*/, 0, 0, 0, 0 /*pod (anti-)affinity*/, 0 /*volumes*/};
  const uint32_t spec_off[2] = {0, (uint32_t)(sizeof words / sizeof words[0])};
  const uint32_t uid[1] = {2};
  const int64_t ts[1] = {0};
  ksh_pod_block blk; memset(&blk, 0, sizeof blk);
  blk.n_pods = 1; blk.n_strings = 3; blk.str_off = str_off; blk.str_bytes = strs; blk.spec_off = spec_off; blk.spec_words = words; blk.uid = uid; blk.creation_ts = ts; blk.str_bytes_len = sizeof strs - 1; blk.spec_words_len = sizeof words / sizeof words[0];
  void* batch = NULL; double ingest_ms = 0;
  if (ksh_pods_ingest(&blk, 1, &batch, &ingest_ms) != KS_OK) { fprintf(stderr, "ingest: %s\n", ksh_last_error()); return 1; }
  uint32_t np = 0, ns = 0; ksh_pods_count(batch, &np, &ns);

  void* h = NULL;
  if (ksh_open_batch(env, batch, 0, &h) != KS_OK) { fprintf(stderr, "flatten: %s\n", ksh_last_error()); return 1; }
  uint32_t dims[10]; ksh_dims(h, dims);
  const ks_problem* p = ksh_problem(h);

  /* the same environment through ITS binary door (argv[2]: n_strings, n_words, str_off[n_strings + 1], words[n_words], string bytes -- what a shim builds in memory):
     the flat problem over it must be the one over the parsed text, array for array */
  if (argc > 2) {
    FILE* g = fopen(argv[2], "rb"); if (!g) return 2;
    uint32_t hd[2]; if (fread(hd, 4, 2, g) != 2) return 2;
    uint32_t* so = (uint32_t*)malloc(((size_t)hd[0] + 1) * 4); uint32_t* wd = (uint32_t*)malloc(((size_t)hd[1] + 1) * 4);
    if (fread(so, 4, (size_t)hd[0] + 1, g) != (size_t)hd[0] + 1 || fread(wd, 4, hd[1], g) != hd[1]) return 2;
    char* sb = (char*)malloc((size_t)so[hd[0]] + 1); if (fread(sb, 1, so[hd[0]], g) != so[hd[0]]) return 2; fclose(g);
    ksh_env_block eb; memset(&eb, 0, sizeof eb);
    eb.n_strings = hd[0]; eb.n_words = hd[1]; eb.str_off = so; eb.str_bytes = sb; eb.words = wd; eb.str_bytes_len = so[hd[0]];
    void* env2 = NULL; double env_ms = 0; void* h2 = NULL;
    if (ksh_env_ingest(&eb, &env2, &env_ms) != KS_OK) { fprintf(stderr, "env ingest: %s\n", ksh_last_error()); return 1; }
    if (ksh_open_batch(env2, batch, 0, &h2) != KS_OK) { fprintf(stderr, "flatten over the binary environment: %s\n", ksh_last_error()); return 1; }
    printf("binary environment: %s flat problem\n", ksh_fingerprint(h2) == ksh_fingerprint(h) ? "the same" : "ANOTHER");
    if (ksh_fingerprint(h2) != ksh_fingerprint(h)) return 1;
    ksh_close(h2); ksh_parsed_free(env2); free(so); free(wd); free(sb);
  }
  printf("pods %u specs %u flat P=%u C=%u T=%u K=%u devices %d\n", np, ns, dims[0], dims[1], p->T, p->K, ks_device_count());

  double ms[6]; void* solved = NULL;
  int rc = ksh_solve_from_batch(env, batch, 0, 0, &solved, ms);
  if (rc == KS_OK) {
    char* out = NULL; ksh_result_text(solved, &out); printf("solved in %.2f ms\n%.60s...\n", ms[5], out); ksh_free(out);
    /* ... and the same result as arrays (no text): every node's pods in commit order, its InstanceTypeOptions, requests and requirement records */
    ksh_result_arrays ra; int started = 0, why = 0, st2[2];
    if (ksh_result_arrays_get(solved, &ra) == KS_OK) {
      for (uint32_t nd = ra.n_existing; nd < ra.n_existing + ra.n_new; ++nd) {
        const uint32_t j = nd - ra.n_existing; uint32_t ntypes = 0;
        for (uint32_t w = 0; w < ra.types_words; ++w) ntypes += (uint32_t)__builtin_popcountll(ra.node_types[(size_t)j * ra.types_words + w]);
        printf("new node %u: template %d, %u pods (first: pod %d), %u instance type options\n", j, ra.node_tmpl[j], ra.node_pods_off[nd + 1] - ra.node_pods_off[nd],
               ra.node_pods_off[nd + 1] > ra.node_pods_off[nd] ? ra.node_pods[ra.node_pods_off[nd]] : -1, ntypes);
        for (uint32_t k = 0; k < ra.n_keys; ++k) if ((ra.node_present[j] >> k) & 1u) printf("  requirement on %s: complement %u, values mask %llx\n", ksh_name(solved, 0, k, 0), (ra.node_complement[j] >> k) & 1u, (unsigned long long)ra.node_mask[(size_t)j * ra.n_keys + k]);
      }
    }
    if (ksh_rr_status(solved, st2) == KS_OK) { started = st2[0]; why = st2[1]; printf("register-resident pack kernel: launched %d, declined with %d\n", started, why); }
    ksh_close(solved);
  }
  else printf("solve refused: %d (%s)\n", rc, ksh_last_error());

  ksh_close(h); ksh_pods_free(batch); ksh_parsed_free(env); free(text);
  return rc == KS_OK || rc == KS_ERR_DEVICE ? 0 : 1;
}
