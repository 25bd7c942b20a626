"""Test harness: a pure (no API server) stand-in for the reference's envtest flow.

`ClusterSim.provision(pods)` plays pkg/test/expectations ExpectProvisioned (expectations.go:215-232):
Solve the batch, "launch" every new node the way the fake cloud provider does (cheapest available
offering, fake/cloudprovider.go:65-114), bind the pods, and keep the resulting nodes/pods as cluster
state for the next batch.  Backends: "oracle" (CPU restatement) and "gpu" (the product: host library
-> C-ABI -> HIP kernels).
"""
from __future__ import annotations

import os
import sys
from collections import Counter
from typing import Dict, List, Optional

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from karpenter_core_amd import fake  # noqa: E402
from karpenter_core_amd.model import (ClusterPod, Container, Expr, LabelSelector, Pod, Problem, Provisioner,  # noqa: E402
                                      SolveResult, StateNode, LABEL_CAPACITY_TYPE, LABEL_HOSTNAME,
                                      LABEL_INSTANCE_TYPE, LABEL_PROVISIONER, LABEL_ZONE, format_milli,
                                      parse_quantity_milli, pod_requests_milli)

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


def solve(problem: Problem, backend: str, **kw) -> SolveResult:
    if backend == "oracle":
        from oracle import oracle_py
        return oracle_py.solve(problem, **kw)
    if backend == "gpu":
        from karpenter_core_amd import scheduler
        return scheduler.solve_problem(problem, **kw)
    raise ValueError(backend)


def selector_matches(sel: Optional[LabelSelector], labels: Dict[str, str]) -> bool:
    if sel is None:
        return False
    for k, v in sel.match_labels.items():
        if labels.get(k) != v:
            return False
    for e in sel.match_expressions:
        has = e.key in labels
        if e.op == "In" and not (has and labels[e.key] in e.values):
            return False
        if e.op == "NotIn" and has and labels[e.key] in e.values:
            return False
        if e.op == "Exists" and not has:
            return False
        if e.op == "DoesNotExist" and has:
            return False
    return True


class ClusterSim:
    def __init__(self, backend: str, instance_types=None, provisioners: Optional[List[Provisioner]] = None,
                 daemonsets: Optional[List[Pod]] = None):
        self.backend = backend
        self.instance_types = instance_types if instance_types is not None else fake.default_instance_types()
        if provisioners is None:
            # scheduling/suite_test.go:94-105 default provisioner; test.Provisioner adds a discovery label and cpu limit 2000
            provisioners = [fake.provisioner("default", len(self.instance_types), limits={"cpu": "2000"},
                                             requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["spot", "on-demand"])],
                                             discovery_label=True)]
        for p in provisioners:
            if not p.instance_types:
                p.instance_types = list(range(len(self.instance_types)))
        self.provisioners = provisioners
        self.daemonsets = daemonsets or []
        self.nodes: List[StateNode] = []
        self.cluster_pods: List[ClusterPod] = []
        self.node_of: Dict[str, str] = {}           # pod uid -> node name
        self.node_types: Dict[str, str] = {}        # node name -> launched instance type
        self.last: Optional[SolveResult] = None
        self._n = 0

    # ---- fake.CloudProvider.Create, cloudprovider.go:65-114 ----
    def _launch(self, nn, prov: Provisioner) -> StateNode:
        def has(req, value):
            if req is None:
                return True
            inside = (value not in req.values) if req.complement else (value in req.values)
            if req.greater_than is not None or req.less_than is not None:
                try:
                    iv = int(value)
                except ValueError:
                    return False
                if req.greater_than is not None and req.greater_than >= iv:
                    return False
                if req.less_than is not None and req.less_than <= iv:
                    return False
            return inside
        zr, cr = nn.requirements.get(LABEL_ZONE), nn.requirements.get(LABEL_CAPACITY_TYPE)
        by_name = {it.name: it for it in self.instance_types}
        best, best_price = None, None
        for name in nn.instance_types:                       # first cheapest wins (canonical for sort.Slice ties)
            it = by_name[name]
            prices = [o.price for o in it.offerings if o.available and has(zr, o.zone) and has(cr, o.capacity_type)]
            if prices and (best_price is None or min(prices) < best_price):
                best, best_price = it, min(prices)
        assert best is not None, "no launchable instance type"
        labels = dict(prov.labels)
        labels[LABEL_PROVISIONER] = prov.name
        for e in best.requirements:
            if e.op == "In" and e.values:
                labels[e.key] = sorted(e.values)[0]
        for o in best.offerings:
            if o.available and has(zr, o.zone) and has(cr, o.capacity_type):
                labels[LABEL_ZONE], labels[LABEL_CAPACITY_TYPE] = o.zone, o.capacity_type
                break
        # a node's labels must satisfy the scheduling node's single-valued requirements
        for k, r in nn.requirements.items():
            if not r.complement and len(r.values) == 1 and k != LABEL_INSTANCE_TYPE:
                labels[k] = r.values[0]
        self._n += 1
        name = f"sim-node-{self._n:04d}"
        labels[LABEL_HOSTNAME] = name
        labels["karpenter.sh/initialized"] = "true"
        alloc = {k: parse_quantity_milli(v) for k, v in best.capacity.items()}
        for k, v in best.overhead.items():
            if k in alloc:
                alloc[k] -= parse_quantity_milli(v)
        self.node_types[name] = best.name
        n = StateNode(name=name, labels=labels, taints=list(prov.taints),
                      available={k: format_milli(v) for k, v in alloc.items()}, capacity=dict(best.capacity))
        n._alloc = alloc
        return n

    def _bind(self, pod: Pod, node: StateNode):
        req = pod_requests_milli(pod)
        avail = {k: parse_quantity_milli(v) for k, v in node.available.items()}
        for k, v in req.items():
            if k in avail:
                avail[k] -= v
        node.available = {k: format_milli(v) for k, v in avail.items()}
        for c in pod.containers:
            node.host_ports.extend(c.ports)
        for v in pod.volumes:                                  # state.Node.updateForPod -> volumeUsage.Add (state/node.go:172)
            if v not in node.volumes:
                node.volumes.append(v)
        self.cluster_pods.append(ClusterPod(uid=pod.uid, namespace=pod.namespace, node_name=node.name,
                                            labels=dict(pod.labels), anti_required=list(pod.anti_required)))
        self.node_of[pod.uid] = node.name

    def problem(self, pods: List[Pod], simulation=False) -> Problem:
        return Problem(instance_types=self.instance_types, provisioners=self.provisioners, pods=pods,
                       daemonset_pods=self.daemonsets, nodes=self.nodes, cluster_pods=self.cluster_pods,
                       extra_well_known=fake.EXTRA_WELL_KNOWN, simulation_mode=simulation)

    def provision(self, pods: List[Pod], bind=True) -> SolveResult:
        res = solve(self.problem(pods), self.backend)
        self.last = res
        if not bind:
            return res
        provs = {p.name: p for p in self.provisioners}
        by_name = {n.name: n for n in self.nodes}
        for name, idxs in res.existing.items():
            for i in idxs:
                self._bind(pods[i], by_name[name])
        for nn in res.new_nodes:
            node = self._launch(nn, provs[nn.provisioner])
            self.nodes.append(node)
            for i in nn.pods:
                self._bind(pods[i], node)
        return res

    # ---- expectations ----
    def scheduled(self, pod: Pod) -> Optional[StateNode]:
        name = self.node_of.get(pod.uid)
        return next((n for n in self.nodes if n.name == name), None) if name else None

    def delete_pod(self, pod: Pod):
        self.cluster_pods = [c for c in self.cluster_pods if c.uid != pod.uid]
        self.node_of.pop(pod.uid, None)

    def skew(self, key: str, selector: LabelSelector, namespace="default") -> List[int]:
        """ExpectSkew, expectations.go:335-360: count of selected, bound pods per domain of `key`."""
        by_name = {n.name: n for n in self.nodes}
        c = Counter()
        for cp in self.cluster_pods:
            if cp.namespace != namespace or not selector_matches(selector, cp.labels):
                continue
            node = by_name[cp.node_name]
            if key == LABEL_HOSTNAME:
                c[node.name] += 1
            elif key in node.labels:
                c[node.labels[key]] += 1
        return sorted(c.values())


_uid = [0]


def mkpod(labels=None, requests=None, **kw) -> Pod:
    _uid[0] += 1
    cont = kw.pop("containers", None) or [Container(requests=dict(requests or {}), limits=dict(kw.pop("limits", {}) or {}),
                                                    ports=list(kw.pop("ports", []) or []))]
    return Pod(uid=kw.pop("uid", f"p{_uid[0]:06d}"), labels=dict(labels or {}), containers=cont, **kw)


def mkpods(n, **kw) -> List[Pod]:
    return [mkpod(**{k: (v.copy() if hasattr(v, "copy") else v) for k, v in kw.items()}) for _ in range(n)]
