// encode.hpp -- flattening of a semantic Solve() problem (ksp::Problem) into the C-ABI `ks_problem`
// (include/ksolve.h) and decoding of `ks_result` back into the reference's result shape.
//
// This is the host half of the boundary: what a Go shim would do inside
// provisioning.(*Provisioner).NewScheduler (provisioner.go:237-296), scheduling.NewTopology
// (topology.go:56-80, countDomains :231-276), scheduling.NewScheduler (scheduler.go:42-94) and
// NewQueue (queue.go:35-41) before handing the flat structure-of-arrays problem to the GPU.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ksolve.h"
#include "hreq.hpp"
#include "ksp.hpp"
#include "kspb.hpp"

namespace ksh {

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

struct ReqSetsStore {
  std::vector<uint32_t> present, complement; std::vector<uint64_t> mask; std::vector<int32_t> gt, lt, it_state;
  uint32_t n = 0;
  ks_reqsets view() const { ks_reqsets v; v.n = n; v.present = present.data(); v.complement = complement.data(); v.mask = mask.data(); v.gt = gt.data(); v.lt = lt.data(); v.it_state = it_state.data(); return v; }
};

struct Encoded {
  std::shared_ptr<const ksp::Problem> src;      // the semantic problem (shared with the caller: a 100k-pod problem is never copied)
  std::shared_ptr<const ksp::PodBatch> batch;   // (binary ingress, kspb.hpp) the pending pods in compact form; `src` then only supplies the environment
  // ---- naming tables for decode ----
  std::vector<std::string> key_names;                      // narrow keys
  std::vector<std::vector<std::string>> key_values;        // universe per key (ascending)
  // A key with more than 64 values that nothing tells apart one by one (only instance types / existing nodes carry them; pods reach the key through
  // Gt / Lt, Exists or a few named values) is encoded over value CLASSES: every named value a class of its own, the others grouped by where they
  // stand relative to every Gt / Lt bound of the problem (and "not an integer").  key_values[k][c] is then the class's representative,
  // key_members[k][c] its values, key_class[k] the value -> class map; for every other key both are empty.
  std::vector<std::vector<std::vector<std::string>>> key_members; std::vector<std::map<std::string, int>> key_class;
  // Integers on a key (label values that parse, Gt / Lt bounds) are compared, never added: when one of them does not fit the kernel's int32 fields the
  // key is encoded over RANKS -- key_ints[k] lists, ascending, every integer the key can meet; value_int and the bounds carry positions in it (Go's int
  // is 64 bits wide: requirement.go:227-269).  Empty for a key whose integers all fit (the encoding is then the integers themselves).
  std::vector<std::vector<long long>> key_ints;
  std::vector<std::string> res_names;
  std::vector<const ksp::Provisioner*> templates;          // weight order
  std::vector<int> existing;                                // indices into src.nodes
  std::vector<Requirement> it_states;                      // instance-type-key requirement per state (index 0 unused)
  // ---- flat storage behind ks_problem ----
  std::vector<uint32_t> key_nvalues; std::vector<int32_t> value_int;
  std::vector<uint32_t> it_present, it_complement; std::vector<uint64_t> it_mask, it_offer; std::vector<double> it_price, it_price_lo; std::vector<int64_t> it_alloc, it_cap;
  std::vector<uint16_t> its_inter; std::vector<uint8_t> its_fail, its_nidne; std::vector<uint64_t> its_types;
  ReqSetsStore tmpl, en, cls, flt;
  std::vector<uint64_t> tmpl_taints, tmpl_types; std::vector<int64_t> tmpl_daemon, tmpl_remaining; std::vector<uint32_t> tmpl_daemon_present, tmpl_limit_present;
  std::vector<uint64_t> en_taints; std::vector<int64_t> en_avail, en_requests; std::vector<uint32_t> en_requests_present, en_port_off;
  std::vector<uint8_t> cls_hn_mode; std::vector<uint32_t> cls_hn_off, hn_list; std::vector<int64_t> cls_requests; std::vector<uint32_t> cls_requests_present;
  std::vector<uint64_t> cls_tolerated; std::vector<uint32_t> cls_port_off; std::vector<uint64_t> ports;
  std::vector<int32_t> en_vol_limit, en_vol_count; std::vector<uint64_t> en_vol_set; std::vector<uint32_t> cls_vol_off, vol_list;   // volume limits of existing nodes
  std::vector<uint32_t> cls_own_off, own_list, cls_sel_off, sel_list, cls_isel_off, isel_list, cls_iown_off, iown_list;
  std::vector<uint32_t> pod_stage_off, stage_cls, queue;
  std::vector<uint8_t> grp_type, grp_active; std::vector<int32_t> grp_key, grp_max_skew, grp_count, grp_hslot, grph_count, grph_extra_pos; std::vector<uint32_t> grp_filter_off;
  ks_problem prob{};
  // A what-if flattened over a shared snapshot (encode_whatif) does not copy the snapshot's catalogue arrays (it_*) nor, when it needs no
  // instance-type state of its own, its lattice tables (its_*, it_states): `shared` keeps the snapshot's flattening alive and these pick it.
  std::shared_ptr<const Encoded> shared; bool shared_lattice = false;
  // (snapshot flattenings only) the same problem resident on a device, per device: what-ifs upload against it (ks_problem_upload_shared)
  mutable std::mutex dev_mu; mutable std::map<int, std::shared_ptr<void>> dev_resident;
  // A what-if DERIVED on the device from the resident snapshot (ks_whatifs_open) has no flattening of its own at all: `view` marks an Encoded that
  // only carries dimensions (prob.P / max_new_nodes / ...) and reads every naming table through `shared`.
  bool view = false;
  const Encoded& names() const { return view && shared ? *shared : *this; }
  const Encoded& catalogue() const { return shared ? *shared : *this; }
  const Encoded& lattice() const { return shared && shared_lattice ? *shared : *this; }

  // ---- result buffers ----
  // Worst-case sized (max_new_nodes rows) but never zero-filled: untouched pages cost nothing, the library writes what it fills.
  template <class T> struct RawBuf { std::unique_ptr<T[]> p; void resize(size_t n) { p.reset(new T[n ? n : 1]); } T* data() const { return p.get(); } };
  struct ResultBuf {
    RawBuf<int32_t> pod_node, pod_stage, pod_seq, unscheduled, node_tmpl, node_gt, node_lt, node_it_state;
    RawBuf<uint64_t> node_types, node_mask; RawBuf<int64_t> node_requests; RawBuf<uint32_t> node_requests_present, node_present, node_complement, pod_reason;
    ks_result r{};
  };
  std::unique_ptr<ResultBuf> make_result() const;
  std::string decode(const ks_result& r, double solve_seconds) const;   // KSR1 text (see model.py parse_result)
};

// What a caller keeps next to the objects it holds (ksh_parse): the flattening of their ENVIRONMENT -- instance types, provisioners, state nodes,
// daemonsets: provisioner.go:237-296 rebuilds all of that per Solve -- for the universe signature of the last batch.  A batch that names the same label
// keys / values / bounds / resources (the steady state of a provisioning loop) adopts it and only flattens its pods.  Thread-safe.
struct EnvBase;
struct EnvCache { std::mutex mu; std::shared_ptr<const EnvBase> base; EnvCache(); ~EnvCache(); };
std::unique_ptr<Encoded> encode(std::shared_ptr<const ksp::Problem> pr, uint32_t flags, EnvCache* cache = nullptr);
std::unique_ptr<Encoded> encode(std::shared_ptr<const ksp::Problem> env, std::shared_ptr<const ksp::PodBatch> batch, uint32_t flags, EnvCache* cache = nullptr);
// Binary pod ingress (include/kshost.h ksh_pods_ingest): blocks of flat pod records -> distinct specs + 16 bytes per pod.
std::shared_ptr<const ksp::PodBatch> ingest_pod_blocks(const ksh_pod_block* blocks, uint32_t n_blocks);
inline std::unique_ptr<Encoded> encode(ksp::Problem&& pr, uint32_t flags) { return encode(std::make_shared<const ksp::Problem>(std::move(pr)), flags); }
uint32_t host_threads();

// Consolidation what-ifs over ONE cluster snapshot (deprovisioning/helpers.go:42-99): the snapshot -- every node a state node, every bound pod
// in the batch -- is flattened once; a what-if (its candidate nodes leave the state nodes, their pods become the pending batch) then only redoes
// what depends on the candidate set: the pod classes / queue / topology groups of ITS pods and remainingResources.  Thread-safe after construction.
struct SnapshotBase;
// `before`: the flattening of the SAME problem object before ksh_env_apply appended nodes / pods to it (nothing else may have changed but the pods' nodes and the
// nodes' available resources / in_state): what does not depend on the events is taken from it when the universes come out the same.  pod_node[i] = -1: bound nowhere.
std::shared_ptr<const SnapshotBase> make_snapshot_base(std::shared_ptr<const ksp::Problem> snapshot, const int32_t* pod_node, uint32_t flags, const SnapshotBase* before = nullptr);
void dispose_later(std::shared_ptr<const void> p);      // destroyed on the library's teardown thread, not on the caller's (a snapshot's flattening: a millisecond of free())
bool snapshot_continued(const SnapshotBase& sb);      // did the flattening take the short road
uint64_t snapshot_fingerprint(const SnapshotBase& sb);      // FNV-1a over the flat problem and the per-node tables behind the device derivation (tests: short road == full run)
std::unique_ptr<Encoded> encode_whatif(const SnapshotBase& sb, const uint32_t* cand, uint32_t ncand, uint32_t flags);
// What deriving what-ifs on the device (include/ksolve.h ks_whatifs_open) needs of a snapshot's flattening.  `eligible`: its what-ifs differ in
// nothing but the pod subset, the removed nodes, remainingResources and -- derived on the device from per-node tables -- which topology groups exist from
// the start and what countDomains finds (no required anti-affinity among bound or cluster pods, at most 64 groups, no volume limits / claims); otherwise `why` says what stands in the way and the what-ifs are flattened one by one.
struct DeltaInputs {
  std::shared_ptr<const Encoded> base; uint32_t n_nodes = 0; const int32_t* node_row = nullptr; const std::vector<std::vector<uint32_t>>* by_node = nullptr;
  const uint32_t* pod_rank = nullptr; const int64_t* node_cap = nullptr /* [n_nodes][R] capacity of a node whose provisioner has limits, masked to the limited resources */;
  const int32_t* node_tmpl = nullptr /* template with limits the node counts against, or -1 */; bool eligible = false; std::string why;
  const ks_whatif_topo* topo = nullptr;      // snapshots with topology groups: the per-node tables a what-if's groups are derived from
};
DeltaInputs delta_inputs(const SnapshotBase& sb);
// CPU self-check (tests): the device derivation of one what-if's topology, restated in plain loops, against that what-if flattened by itself.  "" or the first difference.
std::string check_derived_topology(const SnapshotBase& sb, const uint32_t* cand, uint32_t ncand, uint32_t flags);

}  // namespace ksh
