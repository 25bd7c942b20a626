"""Restatement of the reference's in-memory fake cloud provider catalogue generators
(pkg/cloudprovider/fake/instancetype.go:48-187, cloudprovider.go:116-158) and of the pod/provisioner
fixture shapes of pkg/test (pods.go:62-118, provisioner.go:50-110).  These are fixture *shapes*
re-stated so the parity tests and bench can build the same inputs the reference's suites use; no
reference code is copied.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

from .model import (Expr, InstanceType, Offering, Provisioner, LABEL_ARCH, LABEL_CAPACITY_TYPE,
                    LABEL_INSTANCE_TYPE, LABEL_OS, LABEL_ZONE)

LABEL_INSTANCE_SIZE = "size"            # fake/instancetype.go:34-38
LABEL_EXOTIC = "special"
LABEL_INTEGER = "integer"
RES_GPU_A = "fake.com/vendor-a"
RES_GPU_B = "fake.com/vendor-b"
EXTRA_WELL_KNOWN = [LABEL_INSTANCE_SIZE, LABEL_EXOTIC, LABEL_INTEGER]   # fake/instancetype.go:40-46
DISCOVERY_LABEL = "testing.karpenter.sh/test-id"                       # pkg/test/metadata.go


def _qty_value(q: str) -> float:
    """AsApproximateFloat64 of the handful of quantity forms the catalogue uses."""
    mult = {"Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "k": 1e3, "M": 1e6, "G": 1e9, "T": 1e12, "m": 1e-3}
    for suf in ("Ki", "Mi", "Gi", "Ti", "k", "M", "G", "T", "m"):
        if q.endswith(suf):
            return float(q[: -len(suf)]) * mult[suf]
    return float(q)


def price_from_resources(resources: Dict[str, str]) -> float:
    """priceFromResources, fake/instancetype.go:174-187."""
    price = 0.0
    for k, v in resources.items():
        if k == "cpu":
            price += 0.1 * _qty_value(v)
        elif k == "memory":
            price += 0.1 * _qty_value(v) / 1e9
        elif k in (RES_GPU_A, RES_GPU_B):
            price += 1.0
    return price


def new_instance_type(name: str, resources: Optional[Dict[str, str]] = None, offerings: Optional[List[Offering]] = None,
                      architecture: str = "", operating_systems: Optional[Sequence[str]] = None) -> InstanceType:
    """NewInstanceType, fake/instancetype.go:48-106."""
    res = dict(resources or {})
    if _qty_value(res.get("cpu", "0")) == 0:
        res["cpu"] = "4"
    if _qty_value(res.get("memory", "0")) == 0:
        res["memory"] = "4Gi"
    if _qty_value(res.get("pods", "0")) == 0:
        res["pods"] = "5"
    if not offerings:
        p = price_from_resources(res)
        offerings = [Offering("spot", "test-zone-1", p), Offering("spot", "test-zone-2", p),
                     Offering("on-demand", "test-zone-1", p), Offering("on-demand", "test-zone-2", p),
                     Offering("on-demand", "test-zone-3", p)]
    arch = architecture or "amd64"
    oss = sorted(operating_systems) if operating_systems else sorted(["linux", "windows", "darwin"])
    avail = [o for o in offerings if o.available]
    cpu_int = int(_qty_value(res["cpu"]))          # resource.Quantity.Value() of a whole-core quantity
    large = _qty_value(res["cpu"]) > 4 and _qty_value(res["memory"]) > 8 * 2**30
    reqs = [
        Expr(LABEL_INSTANCE_TYPE, "In", [name]),
        Expr(LABEL_ARCH, "In", [arch]),
        Expr(LABEL_OS, "In", list(oss)),
        Expr(LABEL_ZONE, "In", [o.zone for o in avail]),
        Expr(LABEL_CAPACITY_TYPE, "In", [o.capacity_type for o in avail]),
        Expr(LABEL_INSTANCE_SIZE, "In", ["large"]) if large else Expr(LABEL_INSTANCE_SIZE, "In", ["small"]),
        Expr(LABEL_EXOTIC, "In", ["optional"]) if large else Expr(LABEL_EXOTIC, "DoesNotExist", []),
        Expr(LABEL_INTEGER, "In", [str(cpu_int)]),
    ]
    return InstanceType(name=name, requirements=reqs, offerings=list(offerings), capacity=res,
                        overhead={"cpu": "100m", "memory": "10Mi"})


def instance_types(total: int) -> List[InstanceType]:
    """InstanceTypes(total), fake/instancetype.go:151-164: (i+1) vCPU, 2(i+1) Gi, 10(i+1) pods."""
    return [new_instance_type(f"fake-it-{i}", {"cpu": str(i + 1), "memory": f"{(i + 1) * 2}Gi", "pods": str((i + 1) * 10)})
            for i in range(total)]


def instance_types_assorted() -> List[InstanceType]:
    """InstanceTypesAssorted, fake/instancetype.go:109-143: 7x8x3x2x2x2 = 1344 single-offering types."""
    out = []
    for cpu in (1, 2, 4, 8, 16, 32, 64):
        for mem in (1, 2, 4, 8, 16, 32, 64, 128):
            for zone in ("test-zone-1", "test-zone-2", "test-zone-3"):
                for ct in ("spot", "on-demand"):
                    for os_ in ("linux", "windows"):
                        for arch in ("amd64", "arm64"):
                            res = {"cpu": str(cpu), "memory": f"{mem}Gi"}
                            out.append(new_instance_type(f"{cpu}-cpu-{mem}-mem-{arch}-{os_}-{zone}-{ct}", res,
                                                         [Offering(ct, zone, price_from_resources(res))], arch, [os_]))
    return out


def default_instance_types() -> List[InstanceType]:
    """fake.CloudProvider.GetInstanceTypes default catalogue, fake/cloudprovider.go:116-158."""
    return [
        new_instance_type("default-instance-type"),
        new_instance_type("small-instance-type", {"cpu": "2", "memory": "2Gi"}),
        new_instance_type("gpu-vendor-instance-type", {RES_GPU_A: "2"}),
        new_instance_type("gpu-vendor-b-instance-type", {RES_GPU_B: "2"}),
        new_instance_type("arm-instance-type", {"cpu": "16", "memory": "128Gi"}, architecture="arm64",
                          operating_systems=["ios", "linux", "windows", "darwin"]),
        new_instance_type("single-pod-instance-type", {"pods": "1"}),
    ]


def assorted_ladder(sizes: int, archs: Sequence[str], oss: Sequence[str], zone_sets: Sequence[Sequence[str]],
                    ct_sets: Sequence[Sequence[str]], cpu_step: int = 2) -> List[InstanceType]:
    """Bench catalogue: the InstanceTypes ladder crossed with the
    arch/os/zone/capacity-type variation of InstanceTypesAssorted.  Size i has cpu_step*(i+1) vCPU,
    2x that in Gi and 10 pods per vCPU."""
    out = []
    for i in range(sizes):
        cpu = cpu_step * (i + 1)
        res = {"cpu": str(cpu), "memory": f"{2 * cpu}Gi", "pods": str(10 * cpu)}
        price = price_from_resources(res)
        for arch in archs:
            for os_ in oss:
                for zi, zs in enumerate(zone_sets):
                    for ci, cs in enumerate(ct_sets):
                        offs = [Offering(ct, z, price * (0.7 if ct == "spot" else 1.0)) for z in zs for ct in cs]
                        out.append(new_instance_type(f"l{cpu}-{arch}-{os_}-z{zi}-c{ci}", res, offs, arch, [os_]))
    return out


def provisioner(name: str = "default", instance_type_count: int = 0, weight: int = 0, labels: Optional[Dict[str, str]] = None,
                requirements: Optional[List[Expr]] = None, taints=None, limits: Optional[Dict[str, str]] = None,
                instance_types: Optional[Iterable[int]] = None, discovery_label: bool = False) -> Provisioner:
    """Shape of test.Provisioner (pkg/test/provisioner.go:50-110).  `limits=None` means Spec.Limits nil."""
    lab = dict(labels or {})
    if discovery_label:
        lab[DISCOVERY_LABEL] = "unspecified"
    its = list(instance_types) if instance_types is not None else list(range(instance_type_count))
    return Provisioner(name=name, weight=weight, labels=lab, requirements=list(requirements or []),
                       taints=list(taints or []), limits=limits, instance_types=its)
