"""Binary ENVIRONMENT ingress (include/kshost.h `ksh_env_ingest`, grammar in host/kspb.hpp EnvReader; round 5): instance types + offerings, provisioners, state nodes,
cluster pods, daemonset pods and SimulationMode handed over as one stream of u32 words over one string table instead of KSP1 text (cloudprovider/types.go:72-145,
machinetemplate.go:46-62, state/node.go:61-159, topology.go:231-276).  The flat problem built over it must be the one the text route builds -- every array of it
(`ksh_fingerprint`) -- and malformed blocks must be refused, not read out of bounds."""
import dataclasses

import numpy as np
import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import (ClusterPod, Container, HostPort, LABEL_HOSTNAME, LabelSelector, Pod, PodAffinityTerm, Taint, Volume, env_to_block, pods_to_blocks)
from test_fuzz import fuzz_problem
from test_fuzz_mid import mid_problem


def _problems():
    yield "config3", W.config3(pods=2000, sizes=10, seed=3)
    yield "config2", W.config2(pods=800)
    for seed in (0, 3, 7, 11, 19, 42):
        yield f"fuzz{seed}", fuzz_problem(seed)          # host ports, volumes, existing nodes, limits, taints, two provisioners, daemonsets
    yield "mid4", mid_problem(4)
    its, prov, nodes, bound = W.cluster_snapshot(existing=24, sizes=5, seed=11)
    rs = np.random.RandomState(2)
    for i, pods in enumerate(bound):
        for p in pods:
            if rs.rand() < 0.5:
                p.volumes = [Volume("ebs.csi", f"default/shared-{rs.randint(8)}")]
        nodes[i].volumes = [v for j, v in enumerate(x for p in pods for x in p.volumes) if v not in [y for q in pods for y in q.volumes][:j]]
        nodes[i].volume_limits = {"ebs.csi": int(rs.randint(1, 6))} if rs.rand() < 0.8 else {}
        nodes[i].host_ports = [HostPort(8080 + i, "TCP", "10.0.0.1")] if i % 5 == 0 else []
        nodes[i].taints = [Taint("dedicated", "x", "NoSchedule")] if i % 7 == 0 else []
    pr = W.whatif(its, prov, nodes, bound, [0, 3, 5, 9], True)
    pr.cluster_pods.append(ClusterPod(uid="other", namespace="default", node_name=nodes[1].name, labels={"my-label": "a"},
                                      anti_required=[PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"my-label": "a"}))]))
    pr.daemonset_pods = [Pod(uid="ds-1", creation_ts=17, labels={"k": "v"}, containers=[Container(requests={"cpu": "100m", "memory": "64Mi"})]),
                         Pod(uid="ds-2", node_selector={"kubernetes.io/arch": "arm64"}, containers=[Container(requests={"cpu": "50m"})])]
    pr.simulation_mode = True
    yield "whatif_snapshot", pr


def test_binary_environment_flattens_to_the_same_problem():
    for name, pr in _problems():
        env_text = S.ParsedProblem(dataclasses.replace(pr, pods=[]))
        env_bin = S.ParsedProblem.from_env_block(env_to_block(pr))
        batch = S.PodBatch(pods_to_blocks(pr.pods, 2))
        a, b, c = S.open_batch(env_bin, batch), S.open_batch(env_text, batch), S.FlatProblem(pr)
        try:
            assert a.dims == b.dims == c.dims, name
            assert a.fingerprint() == b.fingerprint() == c.fingerprint(), name
            assert env_bin.ingest_ms >= 0
        finally:
            a.close(); b.close(); c.close(); batch.close(); env_text.close(); env_bin.close()


def test_binary_environment_over_the_small_fuzz_family():
    """64 randomised environments (existing nodes with taints / host ports / volumes, two provisioners with limits and taints, daemonsets): binary == text, array for array."""
    for seed in range(64):
        pr = fuzz_problem(seed)
        env_bin = S.ParsedProblem.from_env_block(env_to_block(pr))
        batch = S.PodBatch(pods_to_blocks(pr.pods, 1))
        a, c = S.open_batch(env_bin, batch), S.FlatProblem(pr)
        try:
            assert a.fingerprint() == c.fingerprint(), seed
        finally:
            a.close(); c.close(); batch.close(); env_bin.close()


def test_the_environment_alone_is_a_valid_problem():
    """An environment without a batch is the text's `PODS 0`: it flattens (ksh_open_parsed) to the same empty Solve."""
    pr = dataclasses.replace(fuzz_problem(7), pods=[])
    a = S.ParsedProblem.from_env_block(env_to_block(pr))
    b = S.ParsedProblem(pr)
    kh = S.libs()[1]
    import ctypes
    ha, hb = ctypes.c_void_p(), ctypes.c_void_p()
    assert kh.ksh_open_parsed(a._p, 0, ctypes.byref(ha)) == 0 and kh.ksh_open_parsed(b._p, 0, ctypes.byref(hb)) == 0
    try:
        assert kh.ksh_fingerprint(ha) == kh.ksh_fingerprint(hb)
    finally:
        kh.ksh_close(ha); kh.ksh_close(hb); a.close(); b.close()


@pytest.mark.parametrize("damage", ["truncated", "string_id", "type_index", "count", "trailing", "offsets", "beyond_the_strings"])
def test_malformed_environment_blocks_are_refused(damage):
    pr = fuzz_problem(3)
    blk = env_to_block(pr)
    w = blk["words"].copy()
    if damage == "truncated":
        blk["words"], blk["n_words"] = w[: len(w) // 2].copy(), len(w) // 2
    elif damage == "string_id":
        w[1 + len(pr.extra_well_known) + 1] = blk["n_strings"] + 7          # the first instance type's name
        blk["words"] = w
    elif damage == "type_index":
        # the last provisioner's last instance-type index sits right before the node count: find it from the back of the provisioner section by rewriting the writer's output
        blk2 = env_to_block(dataclasses.replace(pr, provisioners=[dataclasses.replace(pr.provisioners[0], instance_types=[len(pr.instance_types) + 3])] + list(pr.provisioners[1:])))
        blk = blk2
    elif damage == "count":
        w[0] = 0xFFFFFFF0                                                    # well-known count past the stream
        blk["words"] = w
    elif damage == "trailing":
        blk["words"], blk["n_words"] = np.concatenate([w, np.zeros(3, dtype=np.uint32)]), len(w) + 3
    elif damage == "offsets":
        so = blk["str_off"].copy(); so[2] = so[1] - 1 if so[1] > 0 else 0xFFFFFFFF; blk["str_off"] = so
        if so[2] >= so[1]:
            pytest.skip("cannot build a descending offset here")
    elif damage == "beyond_the_strings":                                      # ascending offsets whose last one lies behind the string bytes the block says it has (round 6: str_bytes_len)
        blk["str_bytes_len"] = int(blk["str_off"][-1]) - 1
    with pytest.raises(S.KSolveError) as e:
        S.ParsedProblem.from_env_block(blk)
    assert e.value.code == S.KS_ERR_INVALID


@pytest.mark.gpu
def test_solve_over_the_binary_environment_is_the_oracles():
    """Environment AND pods through the binary doors, the result through the array door: no text anywhere on the path, and the oracle's result."""
    from oracle import oracle_py as O
    pr = W.config3(pods=3000, sizes=10, seed=5)
    env = S.ParsedProblem.from_env_block(env_to_block(pr))
    batch = S.PodBatch(pods_to_blocks(pr.pods, 4))
    fp, ms = S.solve_from_batch(env, batch, 0)
    try:
        res = fp.result()
        assert res.canonical() == O.solve(pr).canonical()
        ra = fp.result_arrays()
        assert ra["n_new"] == len(res.new_nodes) and int((ra["pod_node"] >= 0).sum()) == len(pr.pods) - len(res.unscheduled)
    finally:
        fp.close(); batch.close(); env.close()
