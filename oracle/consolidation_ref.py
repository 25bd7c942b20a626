"""CPU restatement of the consolidation decision path around simulateScheduling (TEST INFRASTRUCTURE ONLY).

Follows the reference literally, one simulateScheduling (= one oracle Solve) per probe:
  worstLaunchPrice        pkg/controllers/deprovisioning/helpers.go:292-315
  filterByPrice           helpers.go:148-157
  getNodePrices           consolidation.go:277-287
  computeConsolidation    consolidation.go:190-274
  filterOutSameType       multinodeconsolidation.go:132-165
  firstNNodeConsolidationOption  multinodeconsolidation.go:74-114
  SingleNodeConsolidation.ComputeCommand (scan only)  singlenodeconsolidation.go:54-78
  Drift / Expiration .ComputeCommand    drift.go:59-98, expiration.go:68-113
  Validation.ValidateCommand            validation.go:109-172
Requirements are the string-keyed value sets of the oracle's KSR1 output; prices are Python floats (IEEE double,
like Go's float64; the path only compares and sums them in candidate order)."""
import math

from karpenter_core_amd import workloads
from karpenter_core_amd.model import LABEL_CAPACITY_TYPE, LABEL_INSTANCE_TYPE, LABEL_ZONE
from oracle import oracle_py

MAX_FLOAT64 = 1.7976931348623157e308


def req_has(r, value):            # Requirements.Get(key).Has(value); r is None for a missing key (reads as Exists)
    if r is None:
        return True
    if r.complement:
        if value in r.values:
            return False
    elif value not in r.values:
        return False
    if r.greater_than is None and r.less_than is None:
        return True
    try:
        v = int(value)
    except ValueError:
        return False
    if r.greater_than is not None and r.greater_than >= v:
        return False
    if r.less_than is not None and r.less_than <= v:
        return False
    return True


class Narrowed:                  # a requirement replaced by `In [values]` (Requirements.Add of an In requirement onto one that Has them)
    def __init__(self, values):
        self.complement, self.values, self.greater_than, self.less_than = False, tuple(values), None, None


def worst_launch_price(offerings, reqs):
    ofs = [o for o in offerings if o.available]                  # it.Offerings.Available()
    ct, zone = reqs.get(LABEL_CAPACITY_TYPE), reqs.get(LABEL_ZONE)
    if req_has(ct, "spot"):
        spot = [o for o in ofs if o.capacity_type == "spot" and req_has(zone, o.zone)]
        if spot:
            return max(o.price for o in spot)
    if req_has(ct, "on-demand"):
        od = [o for o in ofs if o.capacity_type == "on-demand" and req_has(zone, o.zone)]
        if od:
            return max(o.price for o in od)
    return MAX_FLOAT64


def filter_by_price(types, options, reqs, price):
    return [n for n in options if worst_launch_price(types[n].offerings, reqs) < price]


def offering_get(it, capacity_type, zone):
    for o in it.offerings:
        if o.capacity_type == capacity_type and o.zone == zone:
            return o
    return None


class Cand:
    def __init__(self, snapshot, i):
        lab = snapshot.nodes[i].labels
        self.index, self.name = i, snapshot.nodes[i].name
        self.instance_type, self.capacity_type, self.zone = lab[LABEL_INSTANCE_TYPE], lab[LABEL_CAPACITY_TYPE], lab[LABEL_ZONE]


def get_node_prices(types, cands):
    price = 0.0
    for c in cands:
        o = offering_get(types[c.instance_type], c.capacity_type, c.zone)
        if o is None:
            raise ValueError("unable to determine offering")
        price += o.price
    return price


def canon_reqs(reqs):
    return {k: (r.complement, tuple(sorted(r.values)), r.greater_than, r.less_than) for k, r in reqs.items()}


def compute_consolidation(snapshot, cand_idx, result_sink=None):
    """-> (action, nodes_to_remove, options, requirements dict of requirement-like objects); result_sink (a list) receives the simulation's SolveResult"""
    types = {it.name: it for it in snapshot.instance_types}
    cands = [Cand(snapshot, i) for i in cand_idx]
    # simulateScheduling, helpers.go:42-99: nodes marked for deletion are no state nodes; a candidate that is itself deleting is an error; the batch
    # is the pending pods, then the candidates' pods, then the pods of the deleting nodes
    deleting = [int(j) for j in getattr(snapshot, "deleting", ())]
    if set(cand_idx) & set(deleting):
        raise ValueError("candidate node is deleting")
    problem = workloads.whatif(snapshot.instance_types, snapshot.provisioner, snapshot.nodes, snapshot.bound, list(cand_idx) + deleting)
    problem.pods = list(getattr(snapshot, "pending", [])) + problem.pods
    res = oracle_py.solve(problem)
    if result_sink is not None:
        result_sink.append(res)
    # helpers.go:102-111: `for _, n := range ifn { if n.Node.Labels[LabelNodeInitialized] != "true" { return nil, false, nil } }` -- ifn is every
    # in-state (owned) existing node Solve was given, whether or not it received a pod
    for j, n in enumerate(snapshot.nodes):
        if j not in set(cand_idx) and j not in set(deleting) and n.in_state and n.owned and n.labels.get("karpenter.sh/initialized") != "true":
            return ("do-nothing", [], [], {})
    if res.unscheduled:
        return ("do-nothing", [], [], {})
    if not res.new_nodes:
        return ("delete", [c.name for c in cands], [], {})
    if len(res.new_nodes) != 1:
        return ("do-nothing", [], [], {})
    node = res.new_nodes[0]
    reqs = dict(node.requirements)
    options = filter_by_price(types, node.instance_types, reqs, get_node_prices(types, cands))
    if not options:
        return ("do-nothing", [], [], {})
    all_spot = all(c.capacity_type == "spot" for c in cands)
    if all_spot and req_has(reqs.get(LABEL_CAPACITY_TYPE), "spot"):
        return ("do-nothing", [], [], {})
    ct = reqs.get(LABEL_CAPACITY_TYPE)
    if req_has(ct, "spot") and req_has(ct, "on-demand"):
        reqs[LABEL_CAPACITY_TYPE] = Narrowed(["spot"])
    return ("replace", [c.name for c in cands], options, reqs)


def replacement_command(snapshot, candidates):
    """Drift.ComputeCommand / Expiration.ComputeCommand (drift.go:59-98, expiration.go:68-113) after their candidate filters and sort: the first candidate
    whose simulation can be run decides -- delete if its pods fit the rest of the cluster, otherwise replace it with EVERY node the simulation opened (no
    price stage, any number of nodes; pods left unscheduled are only logged).  -> (action, [node name], [(instance type options, requirements)] per new node)"""
    for i in candidates:
        try:
            sink = []
            compute_consolidation(snapshot, [i], sink)      # (runs simulateScheduling; its own verdict is not used here)
        except ValueError:                                   # errCandidateNodeDeleting: "just retry" with the next candidate
            continue
        res = sink[0]
        name = snapshot.nodes[i].name
        if not res.new_nodes:
            return ("delete", [name], [])
        return ("replace", [name], [(list(n.instance_types), tuple(sorted(canon_reqs(dict(n.requirements)).items()))) for n in res.new_nodes])
    return ("do-nothing", [], [])


def validate_command(snapshot, action, nodes_to_remove, replacement_types, candidates):
    """Validation.ValidateCommand (validation.go:109-172) on the cluster as it is NOW: the command's nodes that are still candidates are simulated again.
    Valid iff every pod schedules and the simulation needs no new node where none was expected, or exactly one whose instance type options contain the
    command's (the simulation applies no price filter, so it may list more)."""
    names = {snapshot.nodes[i].name: i for i in candidates}
    idx = [names[n] for n in nodes_to_remove if n in names]      # mapNodes: the chosen nodes that are still candidates
    if not idx:
        return False
    sink = []
    compute_consolidation(snapshot, idx, sink)                   # (errCandidateNodeDeleting propagates as the error it is in the reference)
    res = sink[0]
    for j, n in enumerate(snapshot.nodes):                       # simulateScheduling's own readiness rule (helpers.go:102-111) -> allPodsScheduled = false
        if j not in set(idx) and j not in set(getattr(snapshot, "deleting", ())) and n.in_state and n.owned and n.labels.get("karpenter.sh/initialized") != "true":
            return False
    if res.unscheduled:
        return False
    if not res.new_nodes:
        return not replacement_types
    if len(res.new_nodes) > 1 or not replacement_types:
        return False
    return instance_types_are_subset(replacement_types, res.new_nodes[0].instance_types)


def filter_out_same_type(snapshot, options, reqs, cand_idx):
    types = {it.name: it for it in snapshot.instance_types}
    existing, prices = set(), {}
    for i in cand_idx:
        c = Cand(snapshot, i)
        existing.add(c.instance_type)
        o = offering_get(types[c.instance_type], c.capacity_type, c.zone)
        if o is None:
            continue
        existing_price = prices.get(c.instance_type, MAX_FLOAT64)
        if o.price < existing_price:
            prices[c.instance_type] = o.price
    max_price = MAX_FLOAT64
    for n in options:
        if n in existing and prices.get(n, 0.0) < max_price:
            max_price = prices.get(n, 0.0)
    return filter_by_price(types, options, reqs, max_price)


def first_n_node_consolidation_option(snapshot, candidates, max_nodes=100):
    if len(candidates) < 2:
        return ("do-nothing", (), (), ())
    lo, hi = 1, max_nodes
    if len(candidates) <= hi:
        hi = len(candidates) - 1
    last = ("do-nothing", [], [], {})
    while lo <= hi:
        mid = (lo + hi) // 2
        prefix = list(candidates[0:mid + 1])
        action, remove, options, reqs = compute_consolidation(snapshot, prefix)
        if action == "replace":
            options = filter_out_same_type(snapshot, options, reqs, prefix)
            if not options:
                action, remove, reqs = "do-nothing", [], {}
        if action in ("replace", "delete"):
            last = (action, remove, options, reqs)
            lo = mid + 1
        else:
            hi = mid - 1
    return canonical(last)


def single_node_consolidation_option(snapshot, candidates):
    for c in candidates:
        try:
            cmd = compute_consolidation(snapshot, [c])
        except ValueError:          # singlenodeconsolidation.go:57-60: log and continue
            continue
        if cmd[0] in ("replace", "delete"):
            return canonical(cmd)
    return ("do-nothing", (), (), ())


def launch_pick(instance_types, node):
    """fake.CloudProvider.Create's instance-type choice (cloudprovider/fake/cloudprovider.go:72-84), literally: keep the options the machine's
    instance-type requirement names (all of them: ToMachine lists exactly the options), order them by the cheapest AVAILABLE offering that the
    zone / capacity-type requirements admit (Offerings.Available().Requirements(reqs).Cheapest(), types.go:126-145) and take the first; ties go to
    the earlier option (sort.Slice leaves them open).  -> (name, cheapest price) or None."""
    by_name = {it.name: it for it in instance_types}
    zr, cr = node.requirements.get(LABEL_ZONE), node.requirements.get(LABEL_CAPACITY_TYPE)
    best = None
    for name in node.instance_types:
        prices = [o.price for o in by_name[name].offerings if o.available and req_has(zr, o.zone) and req_has(cr, o.capacity_type)]
        if prices and (best is None or min(prices) < best[1]):
            best = (name, min(prices))
    return best


def instance_types_are_subset(lhs_names, rhs_names):
    """helpers.go:118-122"""
    return len(set(rhs_names) & set(lhs_names)) == len(set(lhs_names))


def canonical(cmd):
    action, remove, options, reqs = cmd
    return (action, tuple(remove), tuple(options), tuple(sorted(canon_reqs(reqs).items())))
