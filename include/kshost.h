/* kshost.h -- C ABI of libkshost.so: the host half of the MI355X drop-in for karpenter-core's provisioning scheduler.
 *
 * libksolve.so (include/ksolve.h) takes a FLAT problem (structure-of-arrays bitmasks).  libkshost.so is what produces it from the
 * objects a caller holds, i.e. the part of the reference that needs strings and maps and therefore stays on the host:
 *
 *   ksh_parse                 the caller's objects in memory ([]*v1.Pod, []*cloudprovider.InstanceType, []v1alpha5.Provisioner,
 *                             []*state.Node, the cluster pods countDomains would list): KSP1 text -> C++ objects.  (A Go shim would
 *                             build the same objects from its own structs; KSP1 -- grammar in karpenter_core_amd/model.py -- is the
 *                             wire form the Python mirror and the tests use.)
 *   ksh_solve_from_pods       provisioning.(*Provisioner).NewScheduler + scheduling.NewTopology + (*Scheduler).Solve for that pod
 *                             list: provisioner.go:237-296,301-307 / topology.go:56-117 / scheduler.go:42-133 (queue.go:35 NewQueue
 *                             included).  Flatten -> ks_problem_upload -> ks_feasibility_grid -> ks_solve_dev.
 *   ksh_open / ksh_upload / ksh_solve / ksh_solve_batch
 *                             the same in separate steps (flatten once, keep the problem resident in HBM, solve repeatedly / in batches).
 *   ksh_open_whatifs          deprovisioning.simulateScheduling's problem construction for N candidate sets over ONE cluster snapshot
 *                             (helpers.go:42-99): candidates leave the state nodes, their pods become the batch.
 *   ksh_price_filter          filterByPrice / worstLaunchPrice on results still on the device (helpers.go:148-157,292-315).
 *   ksh_result_text / ksh_result_summary
 *                             what callers read from Solve's return values: KSR1 text (Node.Pods, InstanceTypeOptions,
 *                             Requirements, Requests, ExistingNode.Pods, unscheduled queue, relaxation stages), or the fixed-size record
 *                             consolidation needs of a simulation.
 *
 * Conventions: every function returns KS_OK (0) or a negative KS_ERR_* (ksolve.h); ksh_last_error() gives the thread-local message.
 * Handles are opaque; buffers passed in are only read during the call; strings returned through char** are malloc'ed -> ksh_free.
 * There is no CPU scheduling path behind this ABI: without a gfx950 device every solving entry point fails with KS_ERR_DEVICE.
 * Re-entrancy: handles are independent; two threads may solve different handles concurrently (the reference runs the provisioner and
 * the deprovisioner as two goroutines: provisioner.go:102-104, deprovisioning/controller.go:103-105).
 */
#ifndef KSHOST_H
#define KSHOST_H
#include <stddef.h>
#include <stdint.h>

#include "ksolve.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* ksh_last_error(void);
void ksh_free(char* p);

/* ---- objects in memory ----
 * A parsed object also keeps, behind a mutex, what later calls derive from it once: the flattening of its ENVIRONMENT (instance types, provisioners,
 * state nodes, daemonsets -- everything but the pods) for the universe signature of the last batch, which the next batch with the same label keys /
 * values / bounds / resources adopts instead of encoding the catalogue again (KSH_NO_ENV_CACHE=1 turns that off); and, for a cluster snapshot, its
 * own flattening, shared by the what-ifs over it.  It is otherwise immutable: any number of threads may use it at once. */
int ksh_parse(const char* ksp_text, size_t len, void** out_parsed);
void ksh_parsed_free(void* parsed);

/* ---- Solve() for a pod list the caller holds: flatten + upload + HIP kernels + read-back.
 * ms[6] (may be NULL): flatten | upload | static tables + feasibility grid | pack kernel (HIP events) | ks_solve_dev incl. read-back | total.
 * out_handle (may be NULL) receives a ksh_open-style handle holding the problem and its result (ksh_result_text, ksh_solve again, ...). */
int ksh_solve_from_pods(void* parsed, int device, uint32_t flags /* KS_FLAG_* */, void** out_handle, double* ms);

/* ---- binary pod ingress: the pending pods as flat arrays (no text) ----
 * What provisioner.go:301-307 hands to NewScheduler / Solve is a []*v1.Pod.  A cgo shim walks its pods once (one goroutine per block) and
 * fills, per BLOCK: a table of interned strings, per pod one record of u32 words (string ids, counts, int64 milli-quantities as two words;
 * grammar in karpenter_core_amd/host/kspb.hpp, field for field the KSP1 POD record), the uid (a string id) and the creationTimestamp.
 * ksh_pods_ingest copies what it keeps (the buffers are only read during the call -- the cgo pointer rule) and never builds one object
 * per pod: records that are equal word for word share one decoded spec.  The environment -- instance types, provisioners, state nodes,
 * cluster pods, daemonsets -- comes from ksh_parse (of a KSP1 text with `PODS 0`); it changes far less often than the pending batch.
 * ksh_solve_from_batch == ksh_solve_from_pods for that batch: same flat problem (ksh_fingerprint), same result. */
typedef struct ksh_pod_block {
  uint32_t n_pods, n_strings;
  const uint32_t* str_off;      /* [n_strings + 1] byte offsets into str_bytes */
  const char* str_bytes;
  const uint32_t* spec_off;     /* [n_pods + 1] word offsets into spec_words */
  const uint32_t* spec_words;
  const uint32_t* uid;          /* [n_pods] string ids */
  const int64_t* creation_ts;   /* [n_pods] */
  uint64_t str_bytes_len;       /* bytes behind str_bytes, words behind spec_words: a block whose offsets reach beyond them is refused (round 6; ADVICE r05: without them */
  uint64_t spec_words_len;      /* the library could only check that the offsets ascend) */
} ksh_pod_block;
int ksh_pods_ingest(const ksh_pod_block* blocks, uint32_t n_blocks, void** out_batch, double* ms /* ingest time or NULL */);
/* ---- binary ingress for the ENVIRONMENT (round 5): instance types + offerings (cloudprovider/types.go:72-145), provisioners (machinetemplate.go:46-62), state nodes
 * (state/node.go:61-159), the cluster's pods with required anti-affinity (topology.go:231-276), daemonset pods, SimulationMode -- ONE stream of u32 words over ONE string
 * table (grammar: karpenter_core_amd/host/kspb.hpp, EnvReader).  The handle is what ksh_parse returns for the KSP1 text of the same objects with `PODS 0`: the environment
 * argument of ksh_solve_from_batch / ksh_open_batch, freed by ksh_parsed_free.  No text is written or parsed on this path. */
typedef struct ksh_env_block {
  uint32_t n_strings, n_words;
  const uint32_t* str_off;      /* [n_strings + 1] byte offsets into str_bytes */
  const char* str_bytes;
  const uint32_t* words;        /* [n_words] */
  uint64_t str_bytes_len;       /* bytes behind str_bytes: str_off[n_strings] must not reach beyond */
} ksh_env_block;
int ksh_env_ingest(const ksh_env_block* env, void** out_parsed, double* ms /* ingest time or NULL */);
void ksh_pods_free(void* batch);
int ksh_pods_count(void* batch, uint32_t* n_pods, uint32_t* n_specs);
int ksh_solve_from_batch(void* parsed_env, void* batch, int device, uint32_t flags, void** out_handle, double* ms /* as ksh_solve_from_pods */);
int ksh_open_batch(void* parsed_env, void* batch, uint32_t flags, void** out_handle);   /* flatten only (no GPU needed) */
int ksh_open_parsed(void* parsed, uint32_t flags, void** out_handle);                   /* ksh_open for objects already held (ksh_parse) */

/* ---- the same in steps ---- */
int ksh_open(const char* ksp_text, size_t len, uint32_t flags, void** out_handle);      /* parse + flatten (no GPU needed) */
void ksh_close(void* handle);
const ks_problem* ksh_problem(void* handle);                                           /* the flat problem (owned by the handle) */
/* ---- the result as arrays (round 5): what Scheduler.Solve returns (scheduler.go:96: []*Node, []*ExistingNode; Node.Pods / InstanceTypeOptions / Requirements / Requests,
 * machinetemplate.go:77-100 reads them) without the KSR1 text.  Pointers into the handle's result buffers, valid until the handle is solved again or closed.  Node n of the
 * CSR: existing node n (caller's state-node order) for n < n_existing, else new node n - n_existing (creation order); its pods are pod indices (caller's order) in COMMIT
 * order (Node.Pods).  node_types: bit i = the caller's instance type i (the shim keeps the provisioner's own order when it filters: lo.Filter, node.go:138).  Requirement
 * records are masks over a key's interned values: ksh_name(handle, 0, k, 0) names key k, ksh_name(handle, 1, k, v) value v (a key encoded over value classes names one
 * member per class: such keys -- only reachable through Gt / Lt on labels with more than 64 values -- are listed in full by ksh_result_text), ksh_name(handle, 2, r, 0)
 * resource r.  Not for what-ifs derived on the device (their consumer reads ksh_result_summaries' fixed-size records). */
typedef struct ksh_result_arrays {
  uint32_t n_pods, n_existing, n_new, n_unscheduled, types_words, n_resources, n_keys, pad;
  const int32_t* pod_node;                 /* [n_pods] -1 unscheduled | existing node e | n_existing + new node j */
  const int32_t* pod_stage;                /* [n_pods] the relaxation stage the pod ended at (Preferences.Relax, preferences.go:36-56: the shim applies it back to pod.Spec) */
  const uint32_t* pod_reason;              /* [n_pods] ks_result.pod_reason (KS_WHY_* per machine template) for recordSchedulingResults (scheduler.go:135-172) */
  const int32_t* unscheduled;              /* [n_unscheduled] the queue as Solve left it (queue.go:70-72) */
  const uint32_t* node_pods_off;           /* [n_existing + n_new + 1] */
  const int32_t* node_pods;                /* pod indices, node by node, commit order */
  const int32_t* node_tmpl;                /* [n_new] machine template = provisioner in weight order */
  const uint64_t* node_types;              /* [n_new][types_words] InstanceTypeOptions */
  const int64_t* node_requests; const uint32_t* node_requests_present;      /* [n_new][n_resources] milli-units | bit r: the resource is in the list */
  const uint32_t* node_present; const uint32_t* node_complement; const uint64_t* node_mask; const int32_t* node_gt; const int32_t* node_lt;   /* [n_new](*[n_keys]) Requirements after FinalizeScheduling */
  const int32_t* node_it_state;            /* [n_new] the requirement on the instance-type key as a lattice state (0: none; ksh_result_text spells it out) */
} ksh_result_arrays;
int ksh_result_arrays_get(void* handle, ksh_result_arrays* out);
const char* ksh_name(void* handle, int what /* 0 key, 1 value b of key a, 2 resource */, uint32_t a, uint32_t b);
int ksh_rr_status(void* handle, int out[2]);                                           /* ks_problem_rr_status of the handle's device problem: out[0] ks_pack_rr was launched, out[1] why it declined (0: it took the Solve) */
void ksh_dims(void* handle, uint32_t dims[10]);                                        /* P,C,T,M,E,K,R,G,GH,S */
uint64_t ksh_fingerprint(void* handle);                                                /* hash of every array of the flat problem */
int ksh_upload(void* handle, int device);                                              /* idempotent; a second call with another device is an error */
int ksh_upload_batch(void** handles, uint32_t n, int device, uint32_t nthreads);       /* the same for a batch, on host threads (0 = all usable cores) */
/* Solve / grid on a handle that was not uploaded yet use the calling thread's current HIP device. */
int ksh_solve(void* handle, char** out_text /* KSR1 or NULL */, float* kernel_ms, double* wall_ms);
int ksh_solve_batch(void** handles, uint32_t n, char** out_texts /* n entries or NULL */, float* kernel_ms, double* wall_ms);
int ksh_grid(void* handle, uint64_t* out /* [M][C][ceil(T/64)] or NULL */, float* kernel_ms);
/* The grid's rows split over GPUs (ks_feasibility_grid_rows / _install of ksolve.h for a handle): rows [row_lo, row_hi) of the M * C rows computed on the
 * handle's device and copied out (host and / or device destination, either may be NULL); rows computed elsewhere installed (complete != 0 with the last of them). */
int ksh_grid_rows(void* handle, uint32_t row_lo, uint32_t row_hi, uint64_t* out_rows, void* out_rows_dev, float* kernel_ms);
int ksh_grid_install(void* handle, uint32_t row_lo, uint32_t row_hi, const uint64_t* rows, const void* rows_dev, int complete);
int ksh_solve_ksp(const char* ksp_text, size_t len, uint32_t flags, char** out_text);  /* one shot: KSP1 in, KSR1 out */

/* ---- results ---- */
int ksh_result_text(void* handle, char** out_text);
int ksh_result_summary(void* handle, uint64_t* out /* [2 + words]: n_new, n_unscheduled, new node 0's InstanceTypeOptions */, uint32_t words);
int ksh_result_summaries(void** handles, uint32_t n, uint64_t* out /* [n][2 + words] */, uint32_t words);

/* the batched launch with the results LEFT ON THE DEVICE (only error words come back), and the batch's fixed-size records built there into a
 * caller-owned device buffer d_out[n][3 + words] of uint64: [ids[i], n_new, n_unscheduled, new node 0's InstanceTypeOptions] -- the payload of
 * the one all-gather a what-if fan-out needs, with no host hop */
int ksh_solve_batch_resident(void** handles, uint32_t n, float* kernel_ms, double* wall_ms);
int ksh_result_records_dev(void** handles, uint32_t n, const uint64_t* ids, uint32_t words, void* d_out);
/* ... and the fan-out over several GPUs in one call (ks_solve_batch_sharded, include/ksolve.h): shard s = handles[shard_off[s] .. shard_off[s + 1]), all of a shard resident on
 * one device; ks_deal_lpt (ksolve.h) deals what-ifs to shards by predicted work.  out_rows[n][3 + words], ordered by id. */
int ksh_solve_whatifs_sharded(void** handles, const uint32_t* shard_off, uint32_t nshards, const uint64_t* ids, uint32_t words, uint64_t* out_rows, float* kernel_ms_max);

/* ---- consolidation ---- */
int ksh_open_whatifs(const char* snapshot_text, size_t len, uint32_t flags, uint32_t n, const uint32_t* cand_off /* [n+1] */, const uint32_t* cand,
                     const int32_t* pod_node /* node index of every snapshot pod */, uint32_t nthreads /* 0 = all usable cores */, void** out_handles /* [n] */);
int ksh_price_filter(void** handles, uint32_t n, const uint32_t* node, const double* max_price, const uint32_t* spot_only /* or NULL */,
                     uint64_t* out_masks, uint32_t stride_words, uint32_t* out_counts);

/* launch-time pick of the in-memory provider (fake/cloudprovider.go:79-84) / instanceTypesAreSubset (helpers.go:118-122), on device-resident results */
int ksh_launch_pick(void** handles, uint32_t n, const uint32_t* node, int32_t* out_type, int32_t* out_zone, int32_t* out_ct, double* out_price);
const char* ksh_key_value(void* handle, int which /* 0 zone, 1 capacity-type */, int32_t value_id);
int ksh_types_subset(void** handles, uint32_t n, const uint32_t* node, const uint64_t* lhs, uint32_t stride_words, uint32_t* out);
/* What-ifs DERIVED on the device from the resident snapshot (ksolve.h ks_whatifs_open; SURVEY 8b `ks_solve_batch(shared, whatif deltas, ...)`): no
 * per-what-if flattening on the host, an upload of KBs (candidate masks, remainingResources, descriptors); the handles come back already resident on
 * `device`.  Results read exactly like those of ksh_open_whatifs_parsed.  Bound pods may carry spread / affinity / preferred terms: which groups a
 * what-if starts with and what countDomains finds for it follow from per-node tables of the snapshot (ks_whatif_topo).  Required anti-affinity is covered as well: an inverse group
 * exists only while an owner is in the batch or stays bound (decided per what-if on the device; the evaluation skips a group that does not exist), and the
 * staying owners are counted per node.  KS_ERR_UNSUPPORTED -- nothing opened -- where a what-if depends on its candidate set in other ways: more than 1024
 * groups, one spread group shared by pods whose node filters differ, volume limits / claims.  Use ksh_open_whatifs_parsed then. */
int ksh_open_whatifs_derived(void* parsed_snapshot, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand, const int32_t* pod_node, int device, void** out_handles);
int ksh_open_whatifs_parsed(void* parsed_snapshot, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand, const int32_t* pod_node, uint32_t nthreads, void** out_handles);
/* Diagnostic (no GPU needed, not on any solving path): what the device would derive for ONE candidate set -- group activity, domain counts, hostname rows --
 * restated in plain loops over the per-node tables and compared with that what-if flattened by itself.  KS_OK, or KS_ERR_INVALID with the first difference
 * in ksh_last_error(); KS_ERR_UNSUPPORTED for a snapshot ksh_open_whatifs_derived refuses. */
int ksh_check_whatif_derivation(void* parsed_snapshot, uint32_t flags, const uint32_t* cand, uint32_t ncand, const int32_t* pod_node);

/* ---- the snapshot kept current by EVENTS ("cached incremental SoA builder fed from state.Cluster") ----
 * Replaces, for the snapshot consolidation simulates over, what the reference does between two passes of the deprovisioner (deprovisioning/controller.go:64,
 * every 10 s): state.Cluster hears UpdateNode / DeleteNode / UpdatePod / DeletePod (pkg/controllers/state/cluster.go:151-200) and patches its nodes in place
 * (state/node.go:161-182 updateForPod / cleanupForPod; Available() = Allocatable - the requests of the pods bound, node.go:113); the next pass then flattens
 * everything again (helpers.go:42-99 -> provisioner.go:237-296).  Here the events patch the objects ksh_parse holds AND the snapshot's flattening follows them.
 * `ksd_text`:  KSD1 <n>  { NODE+ <KSP1 NODE record without its keyword>  |  NODE- <node name>  |  BIND <node name> POD <KSP1 pod record>  |  UNBIND <pod uid> }*  END
 *   NODE+   a state node joins (slot = the next node index; slots are never reused, candidate sets keep naming nodes by slot)
 *   NODE-   a state node leaves; the pods bound to it are unbound with it
 *   BIND    a pod is bound to a node: it joins the snapshot's pods (index = the next one), the node's available resources shrink by RequestsForPods(pod),
 *           its host ports and volumes join the node's usage
 *   UNBIND  the reverse
 * The first call hands the snapshot's bindings over (`pod_node`, as the what-if calls take it); from then on the library holds them -- every call that takes a
 * `pod_node` accepts NULL for "the library's", ksh_snapshot_bindings reads them.  info[0] = events applied (an event that cannot be applied ends the call with
 * KS_ERR_INVALID; the ones before it stay), info[1] / info[2] = node / pod slots, info[3] = 1 when the snapshot's flattening was CONTINUED from the one before:
 * pods already seen keep their specs, the catalogue's arrays, the universes and the old nodes' requirement rows are taken over (possible while the events bring no
 * label key / value / resource name the universes lack; otherwise, and when no flattening existed yet, the next what-if call flattens from scratch -- same result
 * either way, tests/test_env_apply.py compares the two byte for byte).  The objects are patched in place: do not call while another thread solves over this
 * snapshot; handles opened before the call keep what they were opened with.  KS_ERR_INVALID "spare room ... used up": the snapshot was parsed with room for a
 * quarter more nodes / pods (at least 256 / 4096); ingest it again. */
int ksh_env_apply(void* parsed_snapshot, const int32_t* pod_node /* first call: the bindings; later NULL */, const char* ksd_text, size_t len, uint32_t info[4] /* or NULL */);
int ksh_snapshot_bindings(void* parsed_snapshot, int32_t* out /* [cap] or NULL */, uint32_t cap, uint32_t* n_pods /* or NULL */, uint32_t* n_nodes /* or NULL */);
/* Diagnostic (tests): FNV-1a over the snapshot's flattening -- the flat problem and the per-node tables behind the device derivation; `cold` != 0: of a
 * flattening made from scratch for the comparison (nothing cached is touched). */
int ksh_snapshot_fingerprint(void* parsed_snapshot, const int32_t* pod_node /* or NULL */, uint32_t flags, int cold, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* KSHOST_H */
