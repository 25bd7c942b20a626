"""Label keys with more than 64 values (the reference's own fixture: `fake.InstanceTypes(400)` gives every type its own `integer` label,
instancetype.go:84) used to be refused as soon as a pod referenced the key.  They are now encoded over value CLASSES (host/encode.hpp
`Encoded::key_members`): values nothing names one by one are grouped by where they stand relative to every Gt / Lt bound of the problem.
requirement.go:44-68,227-243: a requirement takes any number of values; Gt / Lt compare the integer value of each."""
import numpy as np
import pytest

from karpenter_core_amd import fake, scheduler as S, workloads as W
from karpenter_core_amd.model import Container, Expr, Pod, Problem

GT = [5, 17, 40, 100, 250]
LT = [30, 120, 333, 390]


def integer_problem(n=600, seed=3, types=400):
    rs = np.random.RandomState(seed)
    its = fake.instance_types(types)
    pods = W.diverse_pods(rs, n)
    for i in range(0, n, 4):
        k = int(rs.randint(5))
        if k == 0:
            ra = [[Expr(fake.LABEL_INTEGER, "Gt", [str(GT[rs.randint(len(GT))])])]]
        elif k == 1:
            ra = [[Expr(fake.LABEL_INTEGER, "Lt", [str(LT[rs.randint(len(LT))])])]]
        elif k == 2:
            ra = [[Expr(fake.LABEL_INTEGER, "Gt", [str(GT[rs.randint(3)])]), Expr(fake.LABEL_INTEGER, "Lt", [str(LT[1 + rs.randint(3)])])]]
        elif k == 3:
            ra = [[Expr(fake.LABEL_INTEGER, "In", [str(v) for v in (7, 64, 200)[: 1 + rs.randint(3)]])]]
        else:
            ra = [[Expr(fake.LABEL_INTEGER, "NotIn", ["2", "3"]), Expr(fake.LABEL_INTEGER, "Gt", ["1"])]]
        pods[i] = Pod(uid=pods[i].uid, labels=pods[i].labels, containers=pods[i].containers, required_affinity=ra)
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)


def test_the_reference_fixture_flattens():
    f = S.FlatProblem(integer_problem())
    assert f.dims["T"] == 400 and f.dims["P"] == 600
    f.close()


def test_too_many_distinguished_values_are_still_refused_loudly():
    rs = np.random.RandomState(1)
    its = fake.instance_types(400)
    pods = [Pod(uid=f"p{i}", containers=[Container(requests={"cpu": "100m"})], required_affinity=[[Expr(fake.LABEL_INTEGER, "In", [str(i + 1)])]]) for i in range(80)]
    with pytest.raises(S.KSolveError) as e:
        S.FlatProblem(Problem(instance_types=its, provisioners=[fake.provisioner("default", 400)], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN))
    assert e.value.code == S.KS_ERR_UNSUPPORTED
    del rs


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5])
def test_gpu_matches_oracle_on_the_reference_fixture(seed):
    from oracle import oracle_py as O
    p = integer_problem(900, seed)
    got = S.solve_problem(p)
    want = O.solve(p)
    assert got.canonical() == want.canonical() and got.reasons == want.reasons
    assert any(fake.LABEL_INTEGER in n.requirements for n in want.new_nodes)


def wide_integer_problem(seed=1):
    """Go's int is 64 bits wide (requirement.go:227-269): label values and Gt / Lt bounds beyond int32 on a custom key."""
    from karpenter_core_amd.model import LABEL_ZONE
    rs = np.random.RandomState(seed)
    its = fake.assorted_ladder(4, ["amd64"], ["linux"], [W.ZONES], [["spot", "on-demand"]])
    big = [5_000_000_000, 7_000_000_000, -9_000_000_000, 12, 2_147_483_647]
    provs = [fake.provisioner(f"p{i}", len(its), weight=10 - i, labels={"example.com/serial": str(v)}) for i, v in enumerate(big)]
    pods = W.diverse_pods(rs, 300)
    for i in range(0, 300, 3):
        k = int(rs.randint(4))
        if k == 0:
            ra = [[Expr("example.com/serial", "Gt", ["6000000000"])]]
        elif k == 1:
            ra = [[Expr("example.com/serial", "Lt", ["-1"])]]
        elif k == 2:
            ra = [[Expr("example.com/serial", "Gt", ["11"]), Expr("example.com/serial", "Lt", ["5000000001"])]]
        else:
            ra = [[Expr("example.com/serial", "Gt", ["9223372036854775000"])]]      # nothing qualifies: the pod stays pending
        pods[i] = Pod(uid=pods[i].uid, labels=pods[i].labels, containers=pods[i].containers, required_affinity=ra)
    del LABEL_ZONE
    return Problem(instance_types=its, provisioners=provs, pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)


def test_integers_beyond_int32_flatten():
    f = S.FlatProblem(wide_integer_problem())
    assert f.dims["M"] == 5
    f.close()


@pytest.mark.gpu
def test_gpu_matches_oracle_with_64_bit_integers():
    from oracle import oracle_py as O
    p = wide_integer_problem()
    got, want = S.solve_problem(p), O.solve(p)
    assert got.canonical() == want.canonical() and got.reasons == want.reasons
    assert want.unscheduled and any(n.requirements.get("example.com/serial") is not None for n in want.new_nodes)
