"""Semantic problem model for one Solve() call, and its KSP1 text serialisation.

This is the *input closure* of the reference hot path -- everything
`provisioning.(*Provisioner).NewScheduler` (reference
pkg/controllers/provisioning/provisioner.go:237-296), `scheduling.NewTopology`
(pkg/controllers/provisioning/scheduling/topology.go:56-80) and
`(*Scheduler).Solve` (scheduler.go:96-133) read: pending pods, daemonset pods,
provisioners (-> MachineTemplates), the instance-type catalogue, the state nodes
and the cluster pods/nodes that `countDomains` (topology.go:231-276) would list
through the API server.

Names follow the Kubernetes / karpenter domain (pods, provisioners, instance
types, offerings, taints, tolerations, topology spread constraints); nothing
here is ML vocabulary.  The objects are plain dataclasses; `Problem.to_ksp()`
writes the line-oriented KSP1 format that both the C++ host library
(karpenter_core_amd/host/ksp.hpp) and the CPU oracle (oracle/) parse.

KSP1 grammar (tokens separated by whitespace; `~` is the empty string; every
list is `<count> item*`):

  quantity : Kubernetes resource.Quantity text ("100m", "1.8G", "10Mi", "4")
  expr     : key op nvals val*        op in In NotIn Exists DoesNotExist Gt Lt
  selector : NIL | SEL nlabels {k v}* nexprs expr*
  term     : topologyKey nns ns* selector          (namespaces pre-resolved by the host,
                                                    topology.go:324-347 needs the API server)
  podspec  : uid namespace creationTs
             L nlabels {k v}*  NS n {k v}*  RA nterms {nexpr expr*}*
             PA nterms {weight nexpr expr*}*  TOL n {key op value effect}*
             C ncont {nreq {res qty}* nlim {res qty}* nports {ip port proto}*}*
             I ninit {nreq {res qty}* nlim {res qty}*}*
             TS n {maxSkew key whenUnsatisfiable selector}*
             AFR n term*  AFP n {weight term}*  ANR n term*  ANP n {weight term}*
             VOL n {driver pvcId}* | VOL -1            (mounted CSI volumes, see Volume; -1 = a lookup failed; the parser accepts records without the section)
  node     : name inState nlabels {k v}* ntaints {key value effect}* available capacity daemonsetRequests
             nports {ip port proto}*  [VL n {driver count}*]  [VU n {driver pvcId}*]      (optional: volume limits / usage)
"""
from __future__ import annotations

import dataclasses
import io
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

# ---- label / taint constants (reference pkg/apis/v1alpha5/labels.go:26-110, k8s.io/api core/v1) ----
LABEL_ZONE = "topology.kubernetes.io/zone"
LABEL_REGION = "topology.kubernetes.io/region"
LABEL_HOSTNAME = "kubernetes.io/hostname"
LABEL_INSTANCE_TYPE = "node.kubernetes.io/instance-type"
LABEL_ARCH = "kubernetes.io/arch"
LABEL_OS = "kubernetes.io/os"
LABEL_CAPACITY_TYPE = "karpenter.sh/capacity-type"
LABEL_PROVISIONER = "karpenter.sh/provisioner-name"
LABEL_INITIALIZED = "karpenter.sh/initialized"

DO_NOT_SCHEDULE = "DoNotSchedule"
SCHEDULE_ANYWAY = "ScheduleAnyway"

NO_SCHEDULE = "NoSchedule"
PREFER_NO_SCHEDULE = "PreferNoSchedule"
NO_EXECUTE = "NoExecute"

RES_CPU = "cpu"
RES_MEMORY = "memory"
RES_PODS = "pods"
RES_EPHEMERAL = "ephemeral-storage"


def _tok(s: str) -> str:
    s = str(s)
    if s == "":
        return "~"
    if any(c.isspace() for c in s) or s == "~":
        raise ValueError(f"KSP1 tokens may not contain whitespace or be '~': {s!r}")
    return s


@dataclass
class Expr:
    """v1.NodeSelectorRequirement / metav1.LabelSelectorRequirement."""
    key: str
    op: str                      # In NotIn Exists DoesNotExist Gt Lt
    values: List[str] = field(default_factory=list)

    def ksp(self, w):
        w.write(f" {_tok(self.key)} {self.op} {len(self.values)}")
        for v in self.values:
            w.write(" " + _tok(v))


@dataclass
class LabelSelector:
    """metav1.LabelSelector; `None` (nil) selects nothing, empty selects everything
    (reference topologygroup.go:246-252 via metav1.LabelSelectorAsSelector)."""
    match_labels: Dict[str, str] = field(default_factory=dict)
    match_expressions: List[Expr] = field(default_factory=list)


def _ksp_selector(sel: Optional[LabelSelector], w):
    if sel is None:
        w.write(" NIL")
        return
    w.write(f" SEL {len(sel.match_labels)}")
    for k in sorted(sel.match_labels):
        w.write(f" {_tok(k)} {_tok(sel.match_labels[k])}")
    w.write(f" {len(sel.match_expressions)}")
    for e in sel.match_expressions:
        e.ksp(w)


@dataclass
class PodAffinityTerm:
    topology_key: str
    label_selector: Optional[LabelSelector] = None
    namespaces: List[str] = field(default_factory=list)   # empty -> the pod's own namespace

    def ksp(self, w):
        w.write(f" {_tok(self.topology_key)} {len(self.namespaces)}")
        for n in self.namespaces:
            w.write(" " + _tok(n))
        _ksp_selector(self.label_selector, w)


@dataclass
class WeightedPodAffinityTerm:
    weight: int
    term: PodAffinityTerm


@dataclass
class TopologySpreadConstraint:
    max_skew: int
    topology_key: str
    when_unsatisfiable: str = DO_NOT_SCHEDULE
    label_selector: Optional[LabelSelector] = None


@dataclass
class Toleration:
    key: str = ""
    operator: str = ""          # "", Equal, Exists
    value: str = ""
    effect: str = ""


@dataclass
class Taint:
    key: str
    value: str = ""
    effect: str = NO_SCHEDULE


@dataclass
class HostPort:
    port: int
    protocol: str = "TCP"
    host_ip: str = ""           # "" -> 0.0.0.0 (reference hostportusage.go:133-136)


@dataclass
class Volume:
    """One mounted volume AFTER the API-server lookups of VolumeUsage.validate (reference pkg/scheduling/volumeusage.go:145-195):
    `driver` is the CSI driver the claim resolves to, `pvc_id` the claim's id.  `resolve_pod_volumes` below does the lookups."""
    driver: str
    pvc_id: str


@dataclass
class Container:
    requests: Dict[str, str] = field(default_factory=dict)
    limits: Dict[str, str] = field(default_factory=dict)
    ports: List[HostPort] = field(default_factory=list)


@dataclass
class PreferredTerm:
    weight: int
    exprs: List[Expr]


@dataclass
class Pod:
    """The v1.Pod fields the path reads (requirements.go:61-78, resources.go:25-119,
    taints.go:28, hostportusage.go:122-144, topology.go:278-322, preferences.go:36-145)."""
    uid: str
    namespace: str = "default"
    creation_ts: int = 0
    labels: Dict[str, str] = field(default_factory=dict)
    node_selector: Dict[str, str] = field(default_factory=dict)
    required_affinity: List[List[Expr]] = field(default_factory=list)      # NodeSelectorTerms (OR)
    preferred_affinity: List[PreferredTerm] = field(default_factory=list)
    tolerations: List[Toleration] = field(default_factory=list)
    containers: List[Container] = field(default_factory=list)
    init_containers: List[Container] = field(default_factory=list)
    spread: List[TopologySpreadConstraint] = field(default_factory=list)
    affinity_required: List[PodAffinityTerm] = field(default_factory=list)
    affinity_preferred: List[WeightedPodAffinityTerm] = field(default_factory=list)
    anti_required: List[PodAffinityTerm] = field(default_factory=list)
    anti_preferred: List[WeightedPodAffinityTerm] = field(default_factory=list)
    volumes: List[Volume] = field(default_factory=list)       # existingnode.go:87-94 (only existing nodes track volumes)
    volume_error: bool = False                                # VolumeUsage.validate returned an error (a claim / class / volume is missing)

    def ksp(self, w):
        w.write(f"{_tok(self.uid)} {_tok(self.namespace)} {int(self.creation_ts)}")
        w.write(f" L {len(self.labels)}")
        for k in sorted(self.labels):
            w.write(f" {_tok(k)} {_tok(self.labels[k])}")
        w.write(f" NS {len(self.node_selector)}")
        for k in sorted(self.node_selector):
            w.write(f" {_tok(k)} {_tok(self.node_selector[k])}")
        w.write(f" RA {len(self.required_affinity)}")
        for term in self.required_affinity:
            w.write(f" {len(term)}")
            for e in term:
                e.ksp(w)
        w.write(f" PA {len(self.preferred_affinity)}")
        for pt in self.preferred_affinity:
            w.write(f" {int(pt.weight)} {len(pt.exprs)}")
            for e in pt.exprs:
                e.ksp(w)
        w.write(f" TOL {len(self.tolerations)}")
        for t in self.tolerations:
            w.write(f" {_tok(t.key)} {_tok(t.operator)} {_tok(t.value)} {_tok(t.effect)}")
        w.write(f" C {len(self.containers)}")
        for c in self.containers:
            _ksp_reslist(c.requests, w)
            _ksp_reslist(c.limits, w)
            w.write(f" {len(c.ports)}")
            for hp in c.ports:
                w.write(f" {_tok(hp.host_ip)} {int(hp.port)} {_tok(hp.protocol)}")
        w.write(f" I {len(self.init_containers)}")
        for c in self.init_containers:
            _ksp_reslist(c.requests, w)
            _ksp_reslist(c.limits, w)
        w.write(f" TS {len(self.spread)}")
        for s in self.spread:
            w.write(f" {int(s.max_skew)} {_tok(s.topology_key)} {s.when_unsatisfiable}")
            _ksp_selector(s.label_selector, w)
        w.write(f" AFR {len(self.affinity_required)}")
        for t in self.affinity_required:
            t.ksp(w)
        w.write(f" AFP {len(self.affinity_preferred)}")
        for wt in self.affinity_preferred:
            w.write(f" {int(wt.weight)}")
            wt.term.ksp(w)
        w.write(f" ANR {len(self.anti_required)}")
        for t in self.anti_required:
            t.ksp(w)
        w.write(f" ANP {len(self.anti_preferred)}")
        for wt in self.anti_preferred:
            w.write(f" {int(wt.weight)}")
            wt.term.ksp(w)
        # (always written, also when empty: the parser tells an optional section from the next record's uid by peeking at a bare token, and a pod may be called "VOL")
        if self.volume_error:
            w.write(" VOL -1")
        else:
            w.write(f" VOL {len(self.volumes)}")
            for v in self.volumes:
                w.write(f" {_tok(v.driver)} {_tok(v.pvc_id)}")


_OPS = {"In": 0, "NotIn": 1, "Exists": 2, "DoesNotExist": 3, "Gt": 4, "Lt": 5}


class PodBlockWriter:
    """Binary pod ingress (include/kshost.h `ksh_pod_block`, grammar in karpenter_core_amd/host/kspb.hpp): what a cgo shim would fill from its
    []*v1.Pod -- a table of interned strings and, per pod, a record of u32 words (string ids, counts, int64 milli-quantities as two words), the
    uid (a string id) and the creationTimestamp.  Maps are written in ascending key order so that equal specs are equal word for word."""

    def __init__(self):
        self._ids: Dict[str, int] = {}
        self._strs: List[bytes] = []
        self._words: List[int] = []
        self._off: List[int] = [0]
        self._uid: List[int] = []
        self._ts: List[int] = []

    def _s(self, s: str) -> int:
        i = self._ids.get(s)
        if i is None:
            i = self._ids[s] = len(self._strs)
            self._strs.append(s.encode())
        return i

    def _map(self, w, m):
        w.append(len(m))
        for k in sorted(m):
            w.append(self._s(k))
            w.append(self._s(m[k]))

    def _res(self, w, rl):
        w.append(len(rl))
        for k in sorted(rl):
            v = parse_quantity_milli(rl[k]) & 0xFFFFFFFFFFFFFFFF
            w.extend((self._s(k), v & 0xFFFFFFFF, v >> 32))

    def _expr(self, w, e):
        w.extend((self._s(e.key), _OPS[e.op], len(e.values)))
        w.extend(self._s(v) for v in e.values)

    def _sel(self, w, sel):
        if sel is None:
            w.append(1)
            return
        w.append(0)
        self._map(w, sel.match_labels)
        w.append(len(sel.match_expressions))
        for e in sel.match_expressions:
            self._expr(w, e)

    def _term(self, w, t):
        w.extend((self._s(t.topology_key), len(t.namespaces)))
        w.extend(self._s(n) for n in t.namespaces)
        self._sel(w, t.label_selector)

    def add(self, p: "Pod"):
        w = self._words
        w.append(self._s(p.namespace))
        self._map(w, p.labels)
        self._map(w, p.node_selector)
        w.append(len(p.required_affinity))
        for term in p.required_affinity:
            w.append(len(term))
            for e in term:
                self._expr(w, e)
        w.append(len(p.preferred_affinity))
        for pt in p.preferred_affinity:
            w.extend((int(pt.weight) & 0xFFFFFFFF, len(pt.exprs)))
            for e in pt.exprs:
                self._expr(w, e)
        w.append(len(p.tolerations))
        for t in p.tolerations:
            w.extend((self._s(t.key), self._s(t.operator), self._s(t.value), self._s(t.effect)))
        w.append(len(p.containers))
        for c in p.containers:
            self._res(w, c.requests)
            self._res(w, c.limits)
            w.append(len(c.ports))
            for hp in c.ports:
                w.extend((self._s(hp.host_ip), int(hp.port) & 0xFFFFFFFF, self._s(hp.protocol)))
        w.append(len(p.init_containers))
        for c in p.init_containers:
            self._res(w, c.requests)
            self._res(w, c.limits)
        w.append(len(p.spread))
        for sp in p.spread:
            w.extend((int(sp.max_skew) & 0xFFFFFFFF, self._s(sp.topology_key), 1 if sp.when_unsatisfiable == SCHEDULE_ANYWAY else 0))
            self._sel(w, sp.label_selector)
        w.append(len(p.affinity_required))
        for t in p.affinity_required:
            self._term(w, t)
        w.append(len(p.affinity_preferred))
        for wt in p.affinity_preferred:
            w.append(int(wt.weight) & 0xFFFFFFFF)
            self._term(w, wt.term)
        w.append(len(p.anti_required))
        for t in p.anti_required:
            self._term(w, t)
        w.append(len(p.anti_preferred))
        for wt in p.anti_preferred:
            w.append(int(wt.weight) & 0xFFFFFFFF)
            self._term(w, wt.term)
        if p.volume_error:
            w.append(0xFFFFFFFF)
        else:
            w.append(len(p.volumes))
            for v in p.volumes:
                w.extend((self._s(v.driver), self._s(v.pvc_id)))
        self._off.append(len(w))
        self._uid.append(self._s(p.uid))
        self._ts.append(int(p.creation_ts))

    def arrays(self) -> dict:
        import numpy as np
        so = np.zeros(len(self._strs) + 1, dtype=np.uint32)
        np.cumsum([len(b) for b in self._strs], out=so[1:])
        return {"n_pods": len(self._uid), "n_strings": len(self._strs), "str_off": so, "str_bytes": np.frombuffer(b"".join(self._strs) + b"\0", dtype=np.uint8).copy(),
                "spec_off": np.asarray(self._off, dtype=np.uint32), "spec_words": np.asarray(self._words if self._words else [0], dtype=np.uint32),
                "uid": np.asarray(self._uid if self._uid else [0], dtype=np.uint32), "creation_ts": np.asarray(self._ts if self._ts else [0], dtype=np.int64)}


def pods_to_blocks(pods: Sequence["Pod"], n_blocks: int = 1) -> List[dict]:
    """The pending pods as `n_blocks` binary blocks (contiguous slices, each with its own string table -- one per filling goroutine in a shim)."""
    n_blocks = max(1, min(n_blocks, max(1, len(pods))))
    out = []
    for b in range(n_blocks):
        w = PodBlockWriter()
        for p in pods[len(pods) * b // n_blocks: len(pods) * (b + 1) // n_blocks]:
            w.add(p)
        out.append(w.arrays())
    return out


class EnvBlockWriter(PodBlockWriter):
    """Binary ENVIRONMENT ingress (include/kshost.h `ksh_env_block`, grammar in karpenter_core_amd/host/kspb.hpp EnvReader): instance types with their offerings,
    provisioners, state nodes, cluster pods, daemonset pods and SimulationMode as one stream of u32 words over one string table -- what a cgo shim would fill from its
    []*cloudprovider.InstanceType / []v1alpha5.Provisioner / []*state.Node instead of printing KSP1 text."""

    def _taints(self, w, ts):
        w.append(len(ts))
        for t in ts:
            w.extend((self._s(t.key), self._s(t.value), self._s(t.effect)))

    def write(self, pr: "Problem") -> dict:
        import struct
        import numpy as np
        w = self._words
        w.append(len(pr.extra_well_known))
        w.extend(self._s(k) for k in pr.extra_well_known)
        w.append(len(pr.instance_types))
        for it in pr.instance_types:
            w.extend((self._s(it.name), len(it.requirements)))
            for e in it.requirements:
                self._expr(w, e)
            w.append(len(it.offerings))
            for o in it.offerings:
                bits = struct.unpack("<Q", struct.pack("<d", float(o.price)))[0]
                w.extend((self._s(o.capacity_type), self._s(o.zone), bits & 0xFFFFFFFF, bits >> 32, 1 if o.available else 0))
            self._res(w, it.capacity)
            self._res(w, it.overhead)
        w.append(len(pr.provisioners))
        for p in pr.provisioners:
            w.extend((self._s(p.name), int(p.weight) & 0xFFFFFFFF))
            self._map(w, p.labels)
            w.append(len(p.requirements))
            for e in p.requirements:
                self._expr(w, e)
            self._taints(w, p.taints)
            w.append(0 if p.limits is None else 1)
            self._res(w, p.limits or {})
            w.append(len(p.instance_types))
            w.extend(int(i) for i in p.instance_types)
        w.append(len(pr.nodes))
        for n in pr.nodes:
            w.extend((self._s(n.name), 1 if n.in_state else 0))
            self._map(w, n.labels)
            self._taints(w, n.taints)
            self._res(w, n.available)
            self._res(w, n.capacity)
            self._res(w, n.daemonset_requests)
            w.append(len(n.host_ports))
            for hp in n.host_ports:
                w.extend((self._s(hp.host_ip), int(hp.port) & 0xFFFFFFFF, self._s(hp.protocol)))
            w.append(len(n.volume_limits))
            for d in sorted(n.volume_limits):
                w.extend((self._s(d), int(n.volume_limits[d]) & 0xFFFFFFFF))
            w.append(len(n.volumes))
            for v in n.volumes:
                w.extend((self._s(v.driver), self._s(v.pvc_id)))
        w.append(len(pr.cluster_pods))
        for cp in pr.cluster_pods:
            w.extend((self._s(cp.uid), self._s(cp.namespace), self._s(cp.node_name)))
            self._map(w, cp.labels)
            w.append(len(cp.anti_required))
            for t in cp.anti_required:
                self._term(w, t)
        w.append(len(pr.daemonset_pods))
        for d in pr.daemonset_pods:
            ts = int(d.creation_ts) & 0xFFFFFFFFFFFFFFFF
            w.extend((self._s(d.uid), ts & 0xFFFFFFFF, ts >> 32, 0))
            at = len(w)
            self.add(d)                       # (the spec record; add() also notes uid / timestamp / offset for a pod block: not used here)
            w[at - 1] = len(w) - at
        w.append(1 if pr.simulation_mode else 0)
        so = np.zeros(len(self._strs) + 1, dtype=np.uint32)
        np.cumsum([len(b) for b in self._strs], out=so[1:])
        return {"n_strings": len(self._strs), "n_words": len(w), "str_off": so, "str_bytes": np.frombuffer(b"".join(self._strs) + b"\0", dtype=np.uint8).copy(),
                "words": np.asarray(w, dtype=np.uint32)}


def env_to_block(pr: "Problem") -> dict:
    """Everything of `pr` but its pending pods as one binary block (`ksh_env_ingest`)."""
    return EnvBlockWriter().write(pr)


def _ksp_reslist(rl: Dict[str, str], w):
    w.write(f" {len(rl)}")
    for k in sorted(rl):
        w.write(f" {_tok(k)} {_tok(rl[k])}")


@dataclass
class Offering:
    """cloudprovider.Offering (reference pkg/cloudprovider/types.go:106-114)."""
    capacity_type: str
    zone: str
    price: float
    available: bool = True


@dataclass
class InstanceType:
    """cloudprovider.InstanceType (types.go:72-89); `overhead` is Overhead.Total() (types.go:100-102)."""
    name: str
    requirements: List[Expr]
    offerings: List[Offering]
    capacity: Dict[str, str]
    overhead: Dict[str, str] = field(default_factory=dict)


@dataclass
class Provisioner:
    """v1alpha5.Provisioner fields read by NewMachineTemplate (machinetemplate.go:46-62),
    OrderByWeight (apis/v1alpha5/provisioner.go:132-136) and NewScheduler (scheduler.go:46-75)."""
    name: str
    weight: int = 0
    labels: Dict[str, str] = field(default_factory=dict)
    requirements: List[Expr] = field(default_factory=list)
    taints: List[Taint] = field(default_factory=list)
    limits: Optional[Dict[str, str]] = None     # None == Spec.Limits nil
    instance_types: List[int] = field(default_factory=list)   # indices into Problem.instance_types


@dataclass
class StateNode:
    """state.Node as the scheduler sees it (state/node.go:61-159): only the *derived* values are
    carried -- Taints() (ephemeral/startup taints already removed), Available(), Capacity(),
    DaemonSetRequests(), HostPortUsage()."""
    name: str
    labels: Dict[str, str] = field(default_factory=dict)
    taints: List[Taint] = field(default_factory=list)
    available: Dict[str, str] = field(default_factory=dict)
    capacity: Dict[str, str] = field(default_factory=dict)
    daemonset_requests: Dict[str, str] = field(default_factory=dict)
    host_ports: List[HostPort] = field(default_factory=list)
    volume_limits: Dict[str, int] = field(default_factory=dict)   # VolumeLimits(): CSINode driver -> Allocatable.Count (state/cluster.go:292-304)
    volumes: List[Volume] = field(default_factory=list)           # VolumeUsage(): volumes of the pods bound to the node
    in_state: bool = True       # passed to NewScheduler as a stateNode (helpers.go:48-61 drops candidates)

    @property
    def owned(self) -> bool:    # state/node.go Owned(): provisioner-name label non-empty
        return self.labels.get(LABEL_PROVISIONER, "") != ""


class VolumeLookupError(Exception):
    """A Get of VolumeUsage.validate failed (volumeusage.go:152-154,175-184)."""


def resolve_pod_volumes(namespace: str, pod_name: str, volume_sources: Sequence[dict], pvcs: Dict[str, dict],
                        storage_classes: Dict[str, str], pvs: Dict[str, Optional[str]]) -> List[Volume]:
    """VolumeUsage.validate (reference pkg/scheduling/volumeusage.go:145-195) over plain dictionaries, for callers (and tests) that hold
    the API objects rather than resolved volumes.
      volume_sources : the pod's Spec.Volumes, each {"name": ..., "pvc": claimName} | {"name": ..., "ephemeral": {"storage_class": s|None,
                       "volume_name": v|""}} | anything else (ignored, :169-171)
      pvcs           : "<namespace>/<claim>" -> {"storage_class": s|None, "volume_name": v|""}
      storage_classes: name -> Provisioner
      pvs            : name -> Spec.CSI.Driver, or None for a non-CSI volume
    Raises VolumeLookupError where the reference returns the Get error (the caller then sets Pod.volume_error)."""
    out: List[Volume] = []
    seen = set()
    for vs in volume_sources:
        if "pvc" in vs:
            pvc_id = f"{namespace}/{vs['pvc']}"
            if pvc_id not in pvcs:
                raise VolumeLookupError(f"persistentvolumeclaim {pvc_id} not found")
            sc, vol = pvcs[pvc_id].get("storage_class"), pvcs[pvc_id].get("volume_name", "")
        elif "ephemeral" in vs:
            pvc_id = f"{namespace}/{pod_name}-{vs['name']}"      # generated claim name, :165
            sc, vol = vs["ephemeral"].get("storage_class"), vs["ephemeral"].get("volume_name", "")
        else:
            continue
        driver = ""
        if vol:                                                   # bound / static: the driver comes from the volume (:176-180)
            if vol not in pvs:
                raise VolumeLookupError(f"persistentvolume {vol} not found")
            driver = pvs[vol] or ""
        elif sc:                                                  # dynamic: from the storage class (:181-186)
            if sc not in storage_classes:
                raise VolumeLookupError(f"storageclass {sc} not found")
            driver = storage_classes[sc]
        if driver and (driver, pvc_id) not in seen:               # volumes is a set per driver (:41-48)
            seen.add((driver, pvc_id))
            out.append(Volume(driver, pvc_id))
    return out


def inject_volume_topology(pod: "Pod", claims: Sequence[str], pvcs: Dict[str, dict], storage_class_topologies: Dict[str, List[List["Expr"]]],
                           pv_node_affinity: Dict[str, Optional[List[List["Expr"]]]]) -> "Pod":
    """VolumeTopology.Inject (reference pkg/controllers/provisioning/volumetopology.go:35-159) over plain dictionaries -- what the provisioner does to a
    pod BEFORE NewScheduler sees it (provisioner.go:195-215): the zones its volumes live in (or may be created in) become node requirements, appended to
    EVERY required node-affinity term so that no relaxation can drop them.
      claims                   : the claim names of the pod's PersistentVolumeClaim volumes, in Spec.Volumes order
      pvcs                     : "<namespace>/<claim>" -> {"storage_class": s|None|"", "volume_name": v|""}
      storage_class_topologies : name -> AllowedTopologies as terms of Expr (only the first term is used, :96-101)
      pv_node_affinity         : name -> Spec.NodeAffinity.Required terms of Expr (first term only, :118-122), or None
    Returns a copy of the pod; raises VolumeLookupError where the reference fails the pod's validation (validatePersistentVolumeClaims :124-159 -- the
    provisioner then leaves the pod out of the batch)."""
    import copy
    added: List[Expr] = []
    for claim in claims:
        pvc_id = f"{pod.namespace}/{claim}"
        if pvc_id not in pvcs:
            raise VolumeLookupError(f"persistentvolumeclaim {pvc_id} not found")
        sc, vol = pvcs[pvc_id].get("storage_class"), pvcs[pvc_id].get("volume_name", "")
        if sc and sc not in storage_class_topologies:
            raise VolumeLookupError(f"storageclass {sc} not found")
        if vol:                                                   # bound: the volume's own node affinity (:70-76)
            if vol not in pv_node_affinity:
                raise VolumeLookupError(f"persistentvolume {vol} not found")
            terms = pv_node_affinity[vol]
            if terms:
                added.extend(copy.deepcopy(terms[0]))
        elif sc:                                                  # to be created: where the storage class may create it (:78-84)
            terms = storage_class_topologies[sc]
            if terms:
                added.extend(Expr(e.key, "In", list(e.values)) for e in terms[0])
    out = copy.deepcopy(pod)
    if not added:
        return out
    if not out.required_affinity:
        out.required_affinity = [[]]
    for term in out.required_affinity:
        term.extend(copy.deepcopy(added))
    return out


TAINT_NODE_NOT_READY = "node.kubernetes.io/not-ready"
TAINT_NODE_UNREACHABLE = "node.kubernetes.io/unreachable"


def state_node_taints(node_taints: Sequence["Taint"], startup_taints: Sequence["Taint"] = (), initialized: bool = False, owned: bool = True) -> List["Taint"]:
    """state.Node.Taints() (reference pkg/controllers/state/node.go:61-78) -- what a caller must put into `StateNode.taints`: the node's taints
    minus the ephemeral not-ready / unreachable NoSchedule taints and, until the node is initialized (and only for nodes we own), minus the
    provisioner's startup taints (a match needs key, value AND effect to agree)."""
    ephemeral = [Taint(TAINT_NODE_NOT_READY, "", NO_SCHEDULE), Taint(TAINT_NODE_UNREACHABLE, "", NO_SCHEDULE)]
    if not initialized and owned:
        ephemeral += list(startup_taints)
    return [t for t in node_taints if not any(e.key == t.key and e.value == t.value and e.effect == t.effect for e in ephemeral)]


def _ksp_node(n: "StateNode", w):
    """The NODE record of KSP1 without its keyword (also the body of a KSD1 `NODE+` event)."""
    w.write(f"{_tok(n.name)} {1 if n.in_state else 0} {len(n.labels)}")
    for k in sorted(n.labels):
        w.write(f" {_tok(k)} {_tok(n.labels[k])}")
    w.write(f" {len(n.taints)}")
    for t in n.taints:
        w.write(f" {_tok(t.key)} {_tok(t.value)} {_tok(t.effect)}")
    _ksp_reslist(n.available, w)
    _ksp_reslist(n.capacity, w)
    _ksp_reslist(n.daemonset_requests, w)
    w.write(f" {len(n.host_ports)}")
    for hp in n.host_ports:
        w.write(f" {_tok(hp.host_ip)} {int(hp.port)} {_tok(hp.protocol)}")
    w.write(f" VL {len(n.volume_limits)}")      # (always written: see Pod.ksp)
    for d in sorted(n.volume_limits):
        w.write(f" {_tok(d)} {int(n.volume_limits[d])}")
    w.write(f" VU {len(n.volumes)}")
    for v in n.volumes:
        w.write(f" {_tok(v.driver)} {_tok(v.pvc_id)}")


def delta_to_ksd(events: Sequence[tuple]) -> str:
    """KSD1 text for `scheduler.ParsedProblem.apply` (kshost.h `ksh_env_apply`): what state.Cluster hears between two passes over the cluster
    (cluster.go UpdateNode / DeleteNode / UpdatePod / DeletePod).  Events, in order:
        ("node+", StateNode) | ("node-", node_name) | ("bind", node_name, Pod) | ("unbind", pod_uid)"""
    w = io.StringIO()
    w.write(f"KSD1 {len(events)}\n")
    for e in events:
        if e[0] == "node+":
            w.write("NODE+ ")
            _ksp_node(e[1], w)
        elif e[0] == "node-":
            w.write(f"NODE- {_tok(e[1])}")
        elif e[0] == "bind":
            w.write(f"BIND {_tok(e[1])} POD ")
            e[2].ksp(w)
        elif e[0] == "unbind":
            w.write(f"UNBIND {_tok(e[1])}")
        else:
            raise ValueError(f"unknown snapshot event {e[0]!r}")
        w.write("\n")
    w.write("END\n")
    return w.getvalue()


@dataclass
class ClusterPod:
    """A pod already bound in the cluster, as seen by countDomains (topology.go:231-276) and
    updateInverseAffinities (topology.go:181-199).  Only countable pods are listed (scheduled,
    not terminal, not terminating -- topology.go:404-406)."""
    uid: str
    namespace: str
    node_name: str
    labels: Dict[str, str] = field(default_factory=dict)
    anti_required: List[PodAffinityTerm] = field(default_factory=list)


@dataclass
class Problem:
    instance_types: List[InstanceType]
    provisioners: List[Provisioner]
    pods: List[Pod]
    daemonset_pods: List[Pod] = field(default_factory=list)
    nodes: List[StateNode] = field(default_factory=list)
    cluster_pods: List[ClusterPod] = field(default_factory=list)
    extra_well_known: List[str] = field(default_factory=list)   # fake provider adds size/special/integer
    simulation_mode: bool = False

    def to_ksp(self) -> str:
        w = io.StringIO()
        w.write("KSP1\n")
        w.write(f"WELLKNOWN {len(self.extra_well_known)}")
        for k in self.extra_well_known:
            w.write(" " + _tok(k))
        w.write("\n")
        w.write(f"ITS {len(self.instance_types)}\n")
        for it in self.instance_types:
            w.write(f"IT {_tok(it.name)} {len(it.requirements)}")
            for e in it.requirements:
                e.ksp(w)
            w.write(f" {len(it.offerings)}")
            for o in it.offerings:
                w.write(f" {_tok(o.capacity_type)} {_tok(o.zone)} {float(o.price)!r} {1 if o.available else 0}")
            _ksp_reslist(it.capacity, w)
            _ksp_reslist(it.overhead, w)
            w.write("\n")
        w.write(f"PROVS {len(self.provisioners)}\n")
        for p in self.provisioners:
            w.write(f"PROV {_tok(p.name)} {int(p.weight)} {len(p.labels)}")
            for k in sorted(p.labels):
                w.write(f" {_tok(k)} {_tok(p.labels[k])}")
            w.write(f" {len(p.requirements)}")
            for e in p.requirements:
                e.ksp(w)
            w.write(f" {len(p.taints)}")
            for t in p.taints:
                w.write(f" {_tok(t.key)} {_tok(t.value)} {_tok(t.effect)}")
            if p.limits is None:
                w.write(" -1")
            else:
                _ksp_reslist(p.limits, w)
            w.write(f" {len(p.instance_types)}")
            for i in p.instance_types:
                w.write(f" {int(i)}")
            w.write("\n")
        w.write(f"NODES {len(self.nodes)}\n")
        for n in self.nodes:
            w.write("NODE ")
            _ksp_node(n, w)
            w.write("\n")
        w.write(f"CPODS {len(self.cluster_pods)}\n")
        for cp in self.cluster_pods:
            w.write(f"CPOD {_tok(cp.uid)} {_tok(cp.namespace)} {_tok(cp.node_name)} {len(cp.labels)}")
            for k in sorted(cp.labels):
                w.write(f" {_tok(k)} {_tok(cp.labels[k])}")
            w.write(f" {len(cp.anti_required)}")
            for t in cp.anti_required:
                t.ksp(w)
            w.write("\n")
        w.write(f"DAEMONS {len(self.daemonset_pods)}\n")
        for p in self.daemonset_pods:
            w.write("POD ")
            p.ksp(w)
            w.write("\n")
        w.write(f"SIM {1 if self.simulation_mode else 0}\n")
        w.write(f"PODS {len(self.pods)}\n")
        for p in self.pods:
            w.write("POD ")
            p.ksp(w)
            w.write("\n")
        w.write("END\n")
        return w.getvalue()


# ---------------------------------------------------------------------------------------------
# Result side: what callers read from Solve()'s return values: Node.Pods,
# Node.InstanceTypeOptions, Node.Requirements, Node.Requests, ExistingNode.Pods.
# ---------------------------------------------------------------------------------------------
@dataclass
class RequirementOut:
    key: str
    complement: bool
    values: Tuple[str, ...]
    greater_than: Optional[int]
    less_than: Optional[int]

    def operator(self) -> str:   # reference requirement.go:186-197
        if self.complement:
            return "NotIn" if self.values else "Exists"
        return "In" if self.values else "DoesNotExist"

    def node_selector_requirement(self) -> Tuple[str, str, Tuple[str, ...]]:
        """Requirement.NodeSelectorRequirement (requirement.go:70-113): the wire form (key, operator, values) that
        `MachineTemplate.ToMachine` puts into `Machine.Spec.Requirements` (machinetemplate.go:77-100).  Bounds win over
        the value set; values come out sorted (`sets.String.List()`)."""
        if self.greater_than is not None:
            return (self.key, "Gt", (str(self.greater_than),))
        if self.less_than is not None:
            return (self.key, "Lt", (str(self.less_than),))
        if self.complement:
            return (self.key, "NotIn", tuple(sorted(self.values))) if self.values else (self.key, "Exists", ())
        return (self.key, "In", tuple(sorted(self.values))) if self.values else (self.key, "DoesNotExist", ())


@dataclass
class NewNodeOut:
    provisioner: str
    pods: List[int]                       # pod indices (into Problem.pods), commit order
    instance_types: List[str]             # order-preserving filter of the provisioner's list
    requests: Dict[str, int]              # milli-units, only keys present in the Go ResourceList
    requirements: Dict[str, RequirementOut]


@dataclass
class SolveResult:
    new_nodes: List[NewNodeOut]
    existing: Dict[str, List[int]]        # state-node name -> pod indices, commit order
    unscheduled: List[int]                # q.List() at exit (queue order)
    final_stage: List[int]                # relaxation stage each pod ended at
    stats: Dict[str, int] = field(default_factory=dict)
    reasons: Dict[int, int] = field(default_factory=dict)   # unscheduled pod -> why its last add() failed: 4 bits per provisioner in weight order
                                                            # (REASON_*; scheduler.go:193-217 collects one error per provisioner)

    def canonical(self) -> dict:
        """Comparable structure (bit-identical parity means these compare equal)."""
        return {
            "new_nodes": [
                {
                    "provisioner": n.provisioner,
                    "pods": list(n.pods),
                    "instance_types": list(n.instance_types),
                    "requests": dict(sorted(n.requests.items())),
                    "requirements": {
                        k: (r.complement, tuple(sorted(r.values)), r.greater_than, r.less_than)
                        for k, r in sorted(n.requirements.items())
                    },
                }
                for n in self.new_nodes
            ],
            "existing": {k: list(v) for k, v in sorted(self.existing.items()) if v},
            "unscheduled": list(self.unscheduled),
            "final_stage": list(self.final_stage),
        }


# Why a provisioner could not take the pod (one 4-bit code per provisioner, weight order): the step of scheduler.add / Node.Add that refused it.
REASON_NONE, REASON_LIMITS, REASON_TAINTS, REASON_HOST_PORTS, REASON_REQUIREMENTS, REASON_TOPOLOGY, REASON_TOPOLOGY_REQUIREMENTS, REASON_NO_INSTANCE_TYPE = range(8)
REASON_TEXT = {REASON_LIMITS: "all available instance types exceed provisioner limits", REASON_TAINTS: "did not tolerate a taint",
               REASON_HOST_PORTS: "host port conflict", REASON_REQUIREMENTS: "incompatible requirements",
               REASON_TOPOLOGY: "unsatisfiable topology constraint", REASON_TOPOLOGY_REQUIREMENTS: "incompatible requirements (topology)",
               REASON_NO_INSTANCE_TYPE: "no instance type satisfied resources and requirements"}


def reason_codes(packed: int, n_provisioners: int) -> List[int]:
    return [(packed >> (4 * m)) & 15 for m in range(min(n_provisioners, 8))]


def parse_result(text: str) -> SolveResult:
    """Parse the KSR1 result text emitted by both the oracle and the host library.

      KSR1
      NEWNODES n
      NODE provisioner npods idx* ntypes name* nreq {res milli}* nrequirements {key c nvals val* gt lt}*
      EXISTING n
      ENODE name npods idx*
      UNSCHEDULED n idx*
      STAGES n stage*
      REASONS n {pod code}*                 (optional)
      STATS n {name value}*
      END
    """
    toks = text.split()
    pos = 0

    def nxt():
        nonlocal pos
        t = toks[pos]
        pos += 1
        return "" if t == "~" else t

    def expect(s):
        t = nxt()
        if t != s:
            raise ValueError(f"KSR1: expected {s!r} got {t!r} at token {pos}")

    expect("KSR1")
    expect("NEWNODES")
    nn = int(nxt())
    new_nodes = []
    for _ in range(nn):
        expect("NODE")
        prov = nxt()
        pods = [int(nxt()) for _ in range(int(nxt()))]
        its = [nxt() for _ in range(int(nxt()))]
        requests = {}
        for _ in range(int(nxt())):
            k = nxt()
            requests[k] = int(nxt())
        reqs = {}
        for _ in range(int(nxt())):
            key = nxt()
            c = nxt() == "1"
            vals = tuple(nxt() for _ in range(int(nxt())))
            gt = nxt()
            lt = nxt()
            reqs[key] = RequirementOut(key, c, vals, None if gt == "-" else int(gt), None if lt == "-" else int(lt))
        new_nodes.append(NewNodeOut(prov, pods, its, requests, reqs))
    expect("EXISTING")
    existing = {}
    for _ in range(int(nxt())):
        expect("ENODE")
        name = nxt()
        existing[name] = [int(nxt()) for _ in range(int(nxt()))]
    expect("UNSCHEDULED")
    unscheduled = [int(nxt()) for _ in range(int(nxt()))]
    expect("STAGES")
    stages = [int(nxt()) for _ in range(int(nxt()))]
    reasons = {}
    if toks[pos] == "REASONS":
        nxt()
        for _ in range(int(nxt())):
            k = int(nxt())
            reasons[k] = int(nxt())
    expect("STATS")
    stats = {}
    for _ in range(int(nxt())):
        k = nxt()
        stats[k] = int(nxt())
    expect("END")
    return SolveResult(new_nodes, existing, unscheduled, stages, stats, reasons)


# ---------------------------------------------------------------------------------------------
# resource.Quantity text -> exact int64 milli-units (mirrors ksp::parse_quantity_milli)
# ---------------------------------------------------------------------------------------------
_SUFFIX = {"": 1000, "m": 1, "k": 10**6, "M": 10**9, "G": 10**12, "T": 10**15, "P": 10**18, "E": 10**21,
           "Ki": 1000 << 10, "Mi": 1000 << 20, "Gi": 1000 << 30, "Ti": 1000 << 40, "Pi": 1000 << 50, "Ei": 1000 << 60}


def parse_quantity_milli(s: str) -> int:
    import re
    from fractions import Fraction
    m = re.fullmatch(r"([+-]?)(\d*)(?:\.(\d*))?((?:[eE][+-]?\d+)|[A-Za-z]*)", s)
    if not m or (m.group(2) == "" and not m.group(3)):
        raise ValueError(f"bad quantity {s!r}")
    sign, ip, fp, suf = m.group(1), m.group(2) or "0", m.group(3) or "", m.group(4)
    mant = Fraction(int(ip + fp), 10 ** len(fp))
    if suf[:1] in ("e", "E") and len(suf) > 1 and suf[1:].lstrip("+-").isdigit():
        mult = Fraction(1000) * Fraction(10) ** int(suf[1:])
    elif suf in _SUFFIX:
        mult = Fraction(_SUFFIX[suf])
    else:
        raise ValueError(f"bad quantity suffix {s!r}")
    v = mant * mult
    if v.denominator != 1:
        raise ValueError(f"quantity finer than 1 milli-unit: {s!r}")
    return -int(v) if sign == "-" else int(v)


def format_milli(v: int) -> str:
    """Exact text form of a milli-unit value."""
    return str(v // 1000) if v % 1000 == 0 else f"{v}m"


def pod_requests_milli(pod: "Pod") -> Dict[str, int]:
    """resources.RequestsForPods(pod) (reference utils/resources/resources.go:25-33,78-119) in milli-units."""
    def merged(c):
        r = {k: parse_quantity_milli(v) for k, v in c.requests.items()}
        for k, v in c.limits.items():
            r.setdefault(k, parse_quantity_milli(v))
        return r
    total: Dict[str, int] = {}
    for c in pod.containers:
        for k, v in merged(c).items():
            total[k] = total.get(k, 0) + v
    for c in pod.init_containers:
        for k, v in merged(c).items():
            if k not in total or v > total[k]:
                total[k] = v
    total["pods"] = 1000
    return total
