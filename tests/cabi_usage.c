/* A plain C99 translation unit that uses both C ABIs the way a cgo shim would: it is compiled and linked by
 * tests/test_cabi.py::test_headers_are_plain_c, and run there without a GPU (every device entry point must then fail loudly with KS_ERR_DEVICE).
 * Nothing here is C++: plain pointers and sizes, no torch types. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ksolve.h"
#include "kshost.h"

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: cabi_usage <file.ksp>\n"); return 2; }
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
  blk.n_pods = 1; blk.n_strings = 3; blk.str_off = str_off; blk.str_bytes = strs; blk.spec_off = spec_off; blk.spec_words = words; blk.uid = uid; blk.creation_ts = ts;
  void* batch = NULL; double ingest_ms = 0;
  if (ksh_pods_ingest(&blk, 1, &batch, &ingest_ms) != KS_OK) { fprintf(stderr, "ingest: %s\n", ksh_last_error()); return 1; }
  uint32_t np = 0, ns = 0; ksh_pods_count(batch, &np, &ns);

  void* h = NULL;
  if (ksh_open_batch(env, batch, 0, &h) != KS_OK) { fprintf(stderr, "flatten: %s\n", ksh_last_error()); return 1; }
  uint32_t dims[10]; ksh_dims(h, dims);
  const ks_problem* p = ksh_problem(h);
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
