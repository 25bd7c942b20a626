/* ksolve.h -- C ABI of libksolve.so: the MI355X (gfx950) drop-in for karpenter-core's provisioning
 * scheduler hot path.
 *
 * What it replaces (reference paths relative to aws/karpenter-core):
 *   ks_solve / ks_solve_dev      <->  (*Scheduler).Solve       pkg/controllers/provisioning/scheduling/scheduler.go:96-133
 *                                     called from provisioner.go:307 and deprovisioning/helpers.go:93
 *   ks_solve_batch               <->  N independent simulateScheduling what-ifs, deprovisioning/helpers.go:42-115
 *                                     (multinodeconsolidation.go:74-114, singlenodeconsolidation.go:43-78)
 *   ks_price_filter_dev          <->  filterByPrice / worstLaunchPrice on a what-if's replacement node,
 *                                     deprovisioning/helpers.go:148-157,292-315 (consolidation.go:238, multinodeconsolidation.go:164)
 *   ks_feasibility_grid          <->  filterInstanceTypesByRequirements for a fresh node, node.go:137-159
 *                                     (compatible && fits && hasOffering over every instance type)
 *   ks_probe_*                   <->  Requirement.Intersection/Has/Operator/Len, Requirements.Compatible
 *                                     pkg/scheduling/requirement.go:117-204, requirements.go:123-206
 *
 * The reference has no FFI for this path (pure Go); a Go shim would flatten
 * []*v1.Pod / []*cloudprovider.InstanceType / []*state.Node into `ks_problem` (INTEGRATION.md shows
 * the cgo stub).  Everything crossing the boundary is plain pointers + sizes: no torch types, no C++
 * types, no callbacks.  Buffers are owned by the caller and only read during the call (cgo pointer
 * rules); results are written into caller-allocated arrays.
 *
 * Encoding (DESIGN.md "Data layout"):
 *   keys      K <= 32 "narrow" label keys, each with a universe of <= 64 values interned in ascending
 *             byte-wise string order (bit i of a mask == value i).  A requirement on key k is
 *             {present, complement, mask, gt, lt} == reference Requirement{complement, values,
 *             greaterThan, lessThan} (requirement.go:36-42).
 *   instance-type key  node-side states x pod-side requirement columns (tables its_inter / its_fail / its_types)
 *             because its universe is the whole catalogue.
 *   hostname key       implicit: new node n owns a placeholder hostname no pod can name (node.go:46);
 *             existing node e owns hostname e.  Pod classes carry {mode, list of existing-node ids}.
 *   resources R <= 8 int64 milli-units; index 0 = cpu, 1 = memory, 2 = pods.
 *   taints    <= 64 distinct (key,value,effect) triples -> u64 masks; tolerations pre-evaluated.
 *   offerings zone x capacity-type pairs (<= 64) -> u64 per instance type.
 *   pods      deduplicated into classes (one per distinct pod spec x relaxation stage); a pod is a
 *             chain of class ids, one per Preferences.Relax stage (preferences.go:36-56).
 *   topology  groups (spread / affinity / anti-affinity, inverse anti-affinity) with pre-counted
 *             domains (topology.go:231-276 needs the API server and therefore stays host-side).
 */
#ifndef KSOLVE_H
#define KSOLVE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KS_MAX_KEYS 32
#define KS_MAX_RES 8
#define KS_MAX_VALUES 64
#define KS_MAX_ITSTATES 65535
#define KS_NO_BOUND_GT INT32_MIN /* "no greaterThan" */
#define KS_NO_BOUND_LT INT32_MAX /* "no lessThan"   */
#define KS_KEY_HOSTNAME (-2)
#define KS_KEY_NONE (-1)

/* error codes (Solve itself never fails: scheduler.go:132 always returns a nil error) */
#define KS_OK 0
#define KS_ERR_INVALID (-1)     /* malformed ks_problem */
#define KS_ERR_UNSUPPORTED (-2) /* feature outside the supported encoding */
#define KS_ERR_DEVICE (-3)      /* HIP failure / no gfx950 device: the library never falls back to a CPU path */
#define KS_ERR_CAPACITY (-4)    /* more new nodes than max_new_nodes */
#define KS_ERR_INTERNAL (-5)    /* device-side watchdog: the pack loop did not terminate within its step bound */

/* A family of requirement sets (reference scheduling.Requirements), n sets x K keys, SoA. */
typedef struct ks_reqsets {
  uint32_t n;
  const uint32_t* present;    /* [n]   bit k: key k has a requirement                         */
  const uint32_t* complement; /* [n]   bit k: requirement.go:38 `complement`                  */
  const uint64_t* mask;       /* [n*K] value set over the key's universe                      */
  const int32_t* gt;          /* [n*K] greaterThan or KS_NO_BOUND_GT                          */
  const int32_t* lt;          /* [n*K] lessThan    or KS_NO_BOUND_LT                          */
  const int32_t* it_state;    /* [n]   state of the instance-type key (0 == key absent)       */
} ks_reqsets;

typedef struct ks_problem {
  /* ---- dimensions ---- */
  uint32_t P;  /* pods in the batch                                   */
  uint32_t C;  /* pod classes                                         */
  uint32_t T;  /* instance types (TW = ceil(T/64) mask words)         */
  uint32_t M;  /* machine templates == provisioners, weight order     */
  uint32_t E;  /* existing (owned, in-state) nodes, caller's order    */
  uint32_t K;  /* narrow keys                                         */
  uint32_t R;  /* resources                                           */
  uint32_t G;  /* topology groups (topologies first, then inverse)    */
  uint32_t GH; /* groups whose key is the hostname                    */
  uint32_t S;  /* instance-type-key states a node can be in (0 = key absent)                       */
  uint32_t SC; /* instance-type-key requirements a pod class / topology filter can carry (0 = none) */
  uint32_t max_new_nodes; /* capacity for scheduling.Node records (<= P is always enough) */
  uint32_t flags;         /* KS_FLAG_* */

  /* ---- keys ---- */
  uint32_t wellknown_mask;     /* bit k: key k in v1alpha5.WellKnownLabels (labels.go:84-92 + provider additions) */
  const uint32_t* key_nvalues; /* [K] */
  const int32_t* value_int;    /* [K*64] strconv.Atoi of the value or INT32_MIN when not an integer (requirement.go:232) */
  int32_t key_zone, key_ct;    /* narrow-key index of topology.kubernetes.io/zone, karpenter.sh/capacity-type, or -1 */
  uint32_t n_ct;               /* offering pair index = zone_value * n_ct + ct_value */

  /* ---- instance types (cloudprovider.InstanceType, types.go:72-89), SoA over T ---- */
  const uint32_t* it_present;    /* [T] */
  const uint32_t* it_complement; /* [T] */
  const uint64_t* it_mask;       /* [K*T]  it_mask[k*T+t] */
  const int64_t* it_alloc;       /* [R*T]  Allocatable() = Capacity - Overhead.Total(), types.go:87-102 */
  const int64_t* it_cap;         /* [R*T]  Capacity (provisioner limits, scheduler.go:273-309) */
  const uint64_t* it_offer;      /* [T]    available (zone,capacity-type) pairs, types.go:106-128 */
  const double* it_price;        /* [T*NP] Offering.Price of available pair p (NP = key_nvalues[key_zone] * n_ct; the highest one if a pair is offered
                                    twice): read by the consolidation price stage only (ks_price_filter_dev); may be NULL */
  const double* it_price_lo;     /* [T*NP] the LOWEST Offering.Price of available pair p (Offerings.Cheapest, types.go:141): read by ks_launch_pick_dev; NULL -> it_price */
  int32_t ct_spot, ct_ondemand;  /* value ids of "spot" / "on-demand" in the capacity-type key's universe, or -1 */
  /* instance-type key lattice */
  const uint16_t* its_inter; /* [S*SC] node state after intersecting node state a with pod-side req b  */
  const uint8_t* its_fail;   /* [S*SC] Requirements.Intersects error for existing=a, incoming=b        */
  const uint8_t* its_nidne;  /* [S]   operator in {NotIn, DoesNotExist}                                */
  const uint64_t* its_types; /* [S*TW] types whose own `instance-type In [name]` passes against state s */

  /* ---- machine templates (machinetemplate.go:46-62), order = OrderByWeight (provisioner.go:132-136) ---- */
  ks_reqsets tmpl;               /* n = M */
  const uint64_t* tmpl_taints;   /* [M] */
  const int64_t* tmpl_daemon;    /* [M*R] getDaemonOverhead, scheduler.go:250-267 */
  const uint32_t* tmpl_daemon_present; /* [M] resource presence bits of that ResourceList */
  const uint64_t* tmpl_types;    /* [M*TW] the provisioner's instance types */
  const uint32_t* tmpl_limit_present;  /* [M] bit r: resource r is limited; 0xFFFFFFFF == Spec.Limits nil (scheduler.go:71-75) */
  const int64_t* tmpl_remaining; /* [M*R] remainingResources after existing nodes (scheduler.go:244-246) */

  /* ---- existing nodes (existingnode.go:41-75) ---- */
  ks_reqsets en;               /* n = E : NewLabelRequirements(node.Labels) (hostname implicit) */
  const uint64_t* en_taints;   /* [E] */
  const int64_t* en_avail;     /* [E*R] state.Node.Available() */
  const int64_t* en_requests;  /* [E*R] remaining daemon requests, clamped at 0 (existingnode.go:44-53) */
  const uint32_t* en_requests_present; /* [E] */
  const uint32_t* en_port_off; /* [E+1] into ports[] */
  /* volume limits (existingnode.go:87-94, volumeusage.go:102-143; new nodes do not track volumes).  ND == 0: no class mounts a limited volume */
  uint32_t ND;                 /* CSI drivers some existing node limits (<= 64) */
  uint32_t SW;                 /* 64-bit words of the shared-claim sets: claims that can be on a node before the pod arrives */
  const int32_t* en_vol_limit; /* [E*ND] VolumeLimits()[driver], INT32_MAX == no limit */
  const int32_t* en_vol_count; /* [E*ND] distinct claims of the driver mounted on the node */
  const uint64_t* en_vol_set;  /* [E*SW] which shared claims those include */

  /* ---- pod classes ---- */
  ks_reqsets cls;                 /* n = C : NewPodRequirements, requirements.go:61-78 */
  const uint8_t* cls_hn_mode;     /* [C] hostname requirement: 0 none, 1 In list, 2 NotIn list (Exists = 2 + empty) */
  const uint32_t* cls_hn_off;     /* [C+1] into hn_list[] (existing-node indices) */
  const uint32_t* hn_list;
  const int64_t* cls_requests;    /* [C*R] resources.RequestsForPods(pod), resources.go:25-33 */
  const uint32_t* cls_requests_present; /* [C] */
  const uint64_t* cls_tolerated;  /* [C] bit i: some toleration ToleratesTaint(taint i) (taints.go:28-40) */
  const uint32_t* cls_port_off;   /* [C+1] into ports[] */
  const uint64_t* ports;          /* proto<<56 | port<<32 | ip_id (ip_id 0 == unspecified 0.0.0.0/::), hostportusage.go:39-57 */
  const uint32_t* cls_vol_off;    /* [C+1] into vol_list[] */
  const uint32_t* vol_list;       /* the class's volumes, ordered by driver: driver<<24 | shared-claim bit        (bit 31 clear)
                                     1<<31 | driver<<24 | n : n claims no other pod or node mounts
                                     0xFFFFFFFF            : VolumeUsage.validate failed -> no existing node accepts the pod */
  /* topology membership of a class (CSR lists of group ids) */
  const uint32_t* cls_own_off;  /* [C+1] groups in Topology.topologies owned by the pod; entry = g | selfSelecting<<31 */
  const uint32_t* own_list;
  const uint32_t* cls_sel_off;  /* [C+1] non-inverse groups whose selector selects the pod (Record, topology.go:120-133) */
  const uint32_t* sel_list;
  const uint32_t* cls_isel_off; /* [C+1] inverse groups selecting the pod (getMatchingTopologies, topology.go:358-362) */
  const uint32_t* isel_list;
  const uint32_t* cls_iown_off; /* [C+1] inverse groups owned by the pod (Record, topology.go:136-141) */
  const uint32_t* iown_list;

  /* ---- pods ---- */
  const uint32_t* pod_stage_off; /* [P+1] into stage_cls[]: one class per relaxation stage */
  const uint32_t* stage_cls;
  const uint32_t* queue;         /* [P] initial queue order: byCPUAndMemoryDescending, queue.go:74-110 */

  /* ---- topology groups (topologygroup.go:53-86) ---- */
  const uint8_t* grp_type;      /* [G] 0 spread, 1 pod affinity, 2 pod anti-affinity */
  const int32_t* grp_key;       /* [G] narrow key index or KS_KEY_HOSTNAME */
  const int32_t* grp_max_skew;  /* [G] */
  const uint8_t* grp_active;    /* [G] 1: exists after NewTopology; 0: created by a later Topology.Update (topology.go:86-117) */
  const uint32_t* grp_filter_off; /* [G+1] TopologyNodeFilter terms (topologynodefilter.go:28-70) into `flt`; empty == always */
  ks_reqsets flt;
  const int32_t* grp_count;     /* [G*64] initial domain counts (countDomains, topology.go:231-276); -1 == not a registered domain */
  const int32_t* grp_hslot;     /* [G] row in the hostname tables or -1 */
  const int32_t* grph_count;    /* [GH*E] initial counts on the existing nodes' hostnames; -1 unregistered */
  const int32_t* grph_extra_pos;/* [GH] hostnames outside the state nodes with count > 0 (pod affinity options) */
  uint32_t n_topologies;        /* groups [0, n_topologies) are Topology.topologies, the rest inverseTopologies */
} ks_problem;

#define KS_FLAG_SIMULATION 1u /* SchedulerOptions.SimulationMode (scheduler.go:37-40); informational */
#define KS_FLAG_STATS 2u      /* also count the reference algorithm's attempts / scanned types (DESIGN.md roofline) */
/* kernel choice PER PROBLEM (round 5; the environment variables KS_NO_RR / KS_ONE_WAVE / KS_NO_LEAN still switch the whole process for A/B runs): for a library
 * two goroutines share -- the provisioner's Solve and a deprovisioner's what-ifs -- a choice has to travel with the problem, not with the process */
#define KS_FLAG_NO_RR 4u      /* do not start ks_pack_rr: ks_pack takes the Solve from the start */
#define KS_FLAG_ONE_WAVE 8u   /* ks_pack's single-wave variant (what a batch runs per what-if) for a single Solve */
#define KS_FLAG_NO_LEAN 16u   /* the general variant on a problem the LEAN one would take */

/* Result of one Solve: caller allocates the arrays (sizes below), the library fills them. */
typedef struct ks_result {
  /* per pod */
  int32_t* pod_node;   /* [P] -1 unscheduled, [0,E) existing node, E+j new node j */
  int32_t* pod_stage;  /* [P] relaxation stage the pod ended at */
  int32_t* pod_seq;    /* [P] commit sequence number (orders Node.Pods), -1 if unscheduled */
  uint32_t* pod_reason; /* [P] 0 if scheduled; else why the LAST scheduler.add failed (scheduler.go:193-217 keeps one error per provisioner):
                           4 bits per machine template m (weight order, m < 8) at bit 4m -- KS_WHY_* -- so a shim can synthesise the
                           reference's "incompatible with provisioner ..., <reason>" messages for recordSchedulingResults (scheduler.go:135-172) */
  /* unscheduled pods in final queue order (q.List(), queue.go:70-72) */
  uint32_t n_unscheduled;
  int32_t* unscheduled; /* [P] */
  /* per new node (creation order) -- what callers read from scheduling.Node */
  uint32_t n_new;
  int32_t* node_tmpl;      /* [max_new_nodes] */
  uint64_t* node_types;    /* [max_new_nodes*TW] InstanceTypeOptions as a bitmask */
  int64_t* node_requests;  /* [max_new_nodes*R] */
  uint32_t* node_requests_present; /* [max_new_nodes] */
  uint32_t* node_present;  /* [max_new_nodes] Requirements after FinalizeScheduling (node.go:111-115) */
  uint32_t* node_complement;
  uint64_t* node_mask;     /* [max_new_nodes*K] */
  int32_t* node_gt;        /* [max_new_nodes*K] */
  int32_t* node_lt;        /* [max_new_nodes*K] */
  int32_t* node_it_state;  /* [max_new_nodes] */
  /* counters */
  uint64_t stats[32];      /* KS_STAT_*; [8..31] are per-phase cycle counters of the pack kernel (tools/phase_profile.py) */
} ks_result;

enum {      /* per-template failure reasons in ks_result.pod_reason */
  KS_WHY_NONE = 0,
  KS_WHY_LIMITS = 1,           /* "all available instance types exceed provisioner limits" (scheduler.go:198-201)              */
  KS_WHY_TAINTS = 2,           /* Taints.Tolerates (node.go:64)                                                                 */
  KS_WHY_HOST_PORTS = 3,       /* HostPortUsage.Validate (node.go:69); cannot happen on a fresh node, kept for completeness     */
  KS_WHY_REQUIREMENTS = 4,     /* nodeRequirements.Compatible(podRequirements) (node.go:77)                                     */
  KS_WHY_TOPOLOGY = 5,         /* Topology.AddRequirements: unsatisfiable topology constraint (node.go:83)                      */
  KS_WHY_TOPOLOGY_REQS = 6,    /* nodeRequirements.Compatible(topologyRequirements) (node.go:87)                                */
  KS_WHY_NO_INSTANCE_TYPE = 7  /* filterInstanceTypesByRequirements left nothing (node.go:94-98)                                 */
};

enum {
  KS_STAT_POPS = 0,        /* queue pops                                             */
  KS_STAT_RELAX = 1,       /* relaxations                                            */
  KS_STAT_FULLCHECKS = 2,  /* candidate nodes that reached the instance-type filter  */
  KS_STAT_FULLFAILS = 3,   /* ... and failed it                                      */
  KS_STAT_REF_ATTEMPTS = 4,/* Node.Add/ExistingNode.Add calls the reference would make (KS_FLAG_STATS) */
  KS_STAT_REF_TYPES = 5,   /* instance types the reference would scan (KS_FLAG_STATS)  */
  KS_STAT_CYCLES = 6,      /* s_memtime ticks spent in the pack kernel (block 0)       */
  KS_STAT_ERR = 7          /* device-side error code (0 ok)                            */
};

/* ---- device-resident problem: upload once, solve many times (inputs resident in HBM) ---- */
typedef struct ks_dev_problem ks_dev_problem;

int ks_device_count(void);                                     /* number of gfx950 devices visible, 0 if none */
int ks_current_device(void);                                   /* the calling thread's current HIP device (hipGetDevice), 0 if none */
int ks_problem_device(const ks_dev_problem* d);                /* device an uploaded problem lives on */
/* Which pack kernel took the last solve of `d`.  A single LEAN Solve is first offered to the register-resident kernel (ks_pack_rr), which DECLINES what it does not
 * cover -- before or during its run, leaving no trace in the result -- whereupon the general kernel (ks_pack) solves it.  *started: ks_pack_rr was launched;
 * *decline_code: 0 it took the Solve, else why it declined (the codes are listed in karpenter_core_amd/csrc/ks_pack_rr.inc; e.g. 1 static limits, 4 more nodes than it
 * holds, 8 its watchdog).  Diagnostics only: the result is the same either way (scheduler.go:96-219). */
int ks_problem_rr_status(const ks_dev_problem* d, int* started, int* decline_code);
int ks_problem_upload(const ks_problem* p, int device, ks_dev_problem** out);
void ks_problem_free(ks_dev_problem* d);
/* Consolidation what-ifs over ONE cluster snapshot (deprovisioning/helpers.go:42-99) differ in their pods and in which state nodes stay, not in
 * the catalogue.  `base` is a resident problem flattened from the same snapshot (ks_problem_prepare'd): arrays of `p` whose HOST pointers are
 * the very arrays `base` was uploaded from (instance types, offerings, prices, the instance-type-key lattice) are not copied again, and the
 * tables derived from the catalogue are shared with it.  `base` must outlive the returned problem.  Anything not shared is uploaded as usual. */
int ks_problem_upload_shared(const ks_problem* p, const ks_dev_problem* base, ks_dev_problem** out);
/* ---- consolidation what-ifs derived ON THE DEVICE from a resident cluster snapshot (SURVEY 8b `ks_solve_batch(shared, whatif deltas, ...)`;
 * deprovisioning/helpers.go:48-61,81-84: a what-if = the snapshot minus its candidate nodes plus their pods).  `base` is the snapshot flattened
 * as ONE problem -- every node an existing node, every bound pod in the batch -- resident with its tables built.  A what-if is then nothing but
 * its candidate set: ks_whatifs_open lays out the state of all n what-ifs in one arena, uploads KBs (candidate masks, remainingResources,
 * descriptors) and builds every batch on the device (the snapshot's queue order restricted to the candidates' pods).  ks_whatifs_problems are
 * ordinary device problems (views owned by the batch) for ks_solve_batch_dev / ks_batch_records_dev / ks_price_filter_dev / ...; in their
 * results pod i is the i-th pod of the what-if in the SNAPSHOT's queue order (ks_whatifs_pod_ids names the snapshot pod behind each) and
 * existing node e is the snapshot's row e (removed nodes receive nothing).  Not for snapshots with volume limits or more than 1024 topology groups
 * (KS_ERR_UNSUPPORTED: the caller flattens those what-ifs one by one). */
typedef struct ks_whatif_batch ks_whatif_batch;
/* Snapshots whose bound pods carry spread / affinity / anti-affinity terms (base G > 0, G <= 1024): what a what-if's topology takes from its
 * candidate set is derived on the device too -- which groups exist from the start (owned by a pod of the batch: topology.go:72-78) and countDomains over
 * the cluster pods that stay (topology.go:231-276) -- from per-node tables of the snapshot: */
typedef struct ks_whatif_topo {
  const int32_t* node_cnt;    /* [G][n_nodes] pods on the node that group g counts when they are NOT in the batch (selector, namespace, node filter, key present);
                                              inverse anti-affinity groups (g >= n_topologies): the bound pods on the node that OWN the group (topology.go:181-199) */
  const int32_t* node_dom;    /* [G][n_nodes] the node's domain for group g: value id of its label on the group's key, -1 if none; hostname-keyed groups: the
                                              pods on the node the group counts that are in NO batch (they count under its hostname even when the node is a candidate) */
  const uint64_t* node_own;   /* [n_nodes][ceil(G/64)] groups (bit g) some pod bound to the node owns at its first relaxation stage.  An inverse group EXISTS in a
                                              what-if only while an owner is in the batch (this bit) or stays bound (a positive count); a value-keyed one that does not
                                              exist is marked for the pack kernel's evaluation to skip, a hostname-keyed one without counts constrains nothing */
  const int32_t* tot;         /* [G][64]      node_cnt summed per domain over every node */
  const int32_t* extra_tot;   /* [GH]         hostname-keyed groups: nodes that are no existing row and count > 0 */
  const int32_t* grph_base;   /* [GH][E]      hostname-keyed groups: what every existing row counts while its node stays; 0 = registered without pods,
                                              -2 = registered only in a group a pod of the batch owns from the start (existingnode.go:73), else unknown (-1) */
} ks_whatif_topo;
int ks_whatifs_open(const ks_dev_problem* base, uint32_t n_nodes, const int32_t* pod_node /* [base P] node of every snapshot pod */,
                    const int32_t* node_row /* [n_nodes] existing-node row in base, -1 if none */, uint32_t n, const uint32_t* cand_off /* [n+1] */,
                    const uint32_t* cand /* node indices */, const uint32_t* n_pods /* [n] pods bound to each candidate set */,
                    const int64_t* remaining /* [n][M][R] remainingResources without the candidates */,
                    const ks_whatif_topo* topo /* NULL for a snapshot without topology groups */, ks_whatif_batch** out);
ks_dev_problem* const* ks_whatifs_problems(ks_whatif_batch* b);
uint32_t ks_whatifs_count(const ks_whatif_batch* b);
int ks_whatifs_pod_ids(ks_whatif_batch* b, uint32_t i, uint32_t* out /* [n_pods of what-if i] snapshot pod ids, what-if pod order */);
void ks_whatifs_free(ks_whatif_batch* b);
int ks_problem_prepare(ks_dev_problem* d);                     /* build the static tables + feasibility grid now (otherwise the first solve does) */
/* Solve on the uploaded problem; kernel time (ms, HIP events on the solve stream) is returned in *kernel_ms if non-NULL. */
int ks_solve_dev(ks_dev_problem* d, ks_result* out, float* kernel_ms);
/* Convenience: upload + solve + free. */
int ks_solve(const ks_problem* p, ks_result* out);
/* N independent problems (consolidation what-ifs): one workgroup each, one launch. */
/* out == NULL: the results stay on the device (only the error words are read back) for ks_batch_records_dev / ks_price_filter_dev / ... */
int ks_solve_batch_dev(ks_dev_problem* const* d, uint32_t n, ks_result* const* out, float* kernel_ms);
/* The fixed-size records a what-if fan-out exchanges (deprovisioning: what consolidation reads of a simulation, consolidation.go:190-260;
 * multinodeconsolidation.go:74-114 probes many candidate sets), built on the device from the results the last ks_solve*_dev left there, into a
 * caller-owned DEVICE buffer d_out[n][3 + words] of uint64: [ids[i], n_new, n_unscheduled, new node 0's InstanceTypeOptions (zero if none)].
 * The buffer is complete when the call returns -- it can be handed to RCCL as is (no host hop). */
int ks_batch_records_dev(ks_dev_problem* const* ds, uint32_t n, const uint64_t* ids, uint32_t words, void* d_out);
/* ---- the what-if fan-out over several GPUs in ONE call (SURVEY 8b `ks_solve_batch(shared, whatifs, n, out, ngpus)`; deprovisioning/helpers.go:42-115 per what-if,
 * multinodeconsolidation.go:74-114 / singlenodeconsolidation.go:54-78 are the callers that would batch them).  The what-ifs are resident as shards, one list of device
 * problems per GPU (every problem of a shard on the same device: ks_whatifs_open over that GPU's copy of the snapshot, or ks_problem_upload).  Every shard is solved in one
 * batched launch on its own device and stream, concurrently (a thread per shard); the fixed-size decision records -- [id, n_new, n_unscheduled, new node 0's
 * InstanceTypeOptions (words)] -- are built on each device and gathered into out_rows[sum shard_n][3 + words], ordered by id.  kernel_ms_max: the slowest shard's launch.
 * ks_deal_lpt decides which shard a what-if goes to from a predicted weight (e.g. its pods): longest first, each to the least loaded shard. */
void ks_deal_lpt(const uint64_t* weight, uint32_t n, uint32_t nshards, uint32_t* shard_of);
int ks_solve_batch_sharded(ks_dev_problem* const* const* shards, const uint32_t* shard_n, const uint64_t* const* shard_ids, uint32_t nshards, uint32_t words,
                           uint64_t* out_rows, float* kernel_ms_max);
int ks_solve_batch(const ks_problem* const* p, uint32_t n, ks_result* const* out);

/* The static pod-class x instance-type feasibility grid for fresh nodes of every template:
 * out_grid[(m*C + c)*TW + w].  Exposed for parity tests and roofline measurement. */
int ks_feasibility_grid(ks_dev_problem* d, uint64_t* out_grid, float* kernel_ms);
/* SURVEY 8e row 2 -- the static grid's rows split over GPUs (node.go:137-159 for a fresh node of template m and a pod of class c is row m * C + c, ceil(T/64) words):
 * ks_feasibility_grid_rows computes rows [row_lo, row_hi) on this device (every other static table in full) and copies them to host memory (out_rows) and / or into
 * device memory of the caller's (out_rows_dev: e.g. its slice of the buffer ONE all-gather fills); ks_feasibility_grid_install puts rows computed elsewhere in place
 * (from host or device memory; complete != 0: every row is in, the problem solves without building its grid again). */
int ks_feasibility_grid_rows(ks_dev_problem* d, uint32_t row_lo, uint32_t row_hi, uint64_t* out_rows, void* out_rows_dev, float* kernel_ms);
int ks_feasibility_grid_install(ks_dev_problem* d, uint32_t row_lo, uint32_t row_hi, const uint64_t* rows, const void* rows_dev, int complete);

/* Consolidation price stage on results that are still on the device (deprovisioning/helpers.go:148-157 filterByPrice over
 * :292-315 worstLaunchPrice; callers consolidation.go:238 and multinodeconsolidation.go:164): for problem i, of new node
 * node[i]'s InstanceTypeOptions (as left by the last ks_solve*_dev of ds[i]) keep the types whose worst launch price under
 * that node's zone / capacity-type requirements is < max_price[i].  out_types[i] receives ceil(T/64) words, out_counts[i]
 * the number of types kept.  spot_only[i] != 0 prices node i as if its capacity-type requirement had already been narrowed
 * to In [spot] (computeConsolidation does that before multi-node consolidation filters again, consolidation.go:262-265);
 * spot_only may be NULL.  One launch for the whole batch. */
int ks_price_filter_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, const double* max_price,
                        const uint32_t* spot_only, uint64_t* const* out_types, uint32_t* out_counts);

/* Launch-time instance-type pick of the reference's in-memory provider (cloudprovider/fake/cloudprovider.go:79-84: order the machine's
 * InstanceTypeOptions by `Offerings.Available().Requirements(reqs).Cheapest().Price`, types.go:126-145, and take the first): for problem i,
 * of new node node[i]'s InstanceTypeOptions (as left by the last ks_solve*_dev) the type whose cheapest available offering under the node's zone /
 * capacity-type requirements is cheapest -- a wave-wide arg-min over the surviving-type mask.  Ties go to the lowest instance-type index (the
 * reference's sort.Slice leaves them to pdqsort).  out_type[i] = -1 when no option has a compatible available offering; out_pair[i] is the
 * (zone, capacity-type) pair of that cheapest offering (zone_value * n_ct + ct_value), out_price[i] its price.  One launch for the batch. */
int ks_launch_pick_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, int32_t* out_type, int32_t* out_pair, double* out_price);

/* instanceTypesAreSubset (deprovisioning/helpers.go:118-122; consolidation.go:177, validation.go:164): is lhs[i] (a host-held type mask of
 * ceil(T/64) words, row stride `stride_words`, e.g. a command's price-filtered replacement options) a subset of new node node[i]'s
 * InstanceTypeOptions on the device?  out[i] = 1 / 0. */
int ks_types_subset_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, const uint64_t* lhs, uint32_t stride_words, uint32_t* out);

/* ---- requirement-algebra probes (one key); the same device functions the kernels use ---- */
typedef struct ks_req1 { uint64_t mask; int32_t gt, lt; uint8_t present, complement; } ks_req1;
/* value_int: [64] integer value of each universe entry or INT32_MIN.  on_device != 0 runs a 1-thread kernel. */
int ks_probe_intersection(const ks_req1* a, const ks_req1* b, const int32_t* value_int, uint32_t nvalues, int on_device, ks_req1* out);
int ks_probe_compatible(const ks_req1* a, const ks_req1* b, int well_known, const int32_t* value_int, uint32_t nvalues, int on_device, int* ok);

/* Requirement.Has over the key's universe (bit v: Has(value v)), Operator() (0 In, 1 NotIn, 2 Exists, 3 DoesNotExist), Len(), and the two
 * predicates of them the kernels branch on: nidne = operator in {NotIn, DoesNotExist}, len0 = Len() == 0 (requirement.go:171-204). */
typedef struct ks_req_facts { uint64_t has_mask; int64_t len; int32_t op; uint8_t nidne, len0; } ks_req_facts;
int ks_probe_has(const ks_req1* a, const int32_t* value_int, uint32_t nvalues, int on_device, ks_req_facts* out);

const char* ks_last_error(void); /* thread-local message of the last non-OK return */
const char* ks_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KSOLVE_H */
