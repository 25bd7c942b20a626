//go:build ks_harness

// solve_dump_test.go -- NOT part of this repository's build or tests (there is no Go toolchain in its image).  It is the harness a maintainer with the reference checked out
// would use to close the parity loop BASELINE.md section 2 leaves open: the same fixture through the REFERENCE's Scheduler.Solve, printed in the line format
// tools/go_harness/export_fixture.py --want prints for this repository's oracle (and, through the same lines, for the GPU result).
//
//   cp tools/go_harness/solve_dump_test.go  <karpenter-core>/pkg/controllers/provisioning/scheduling/
//   python tools/go_harness/export_fixture.py config3 --pods 20000 > /tmp/fixture.json
//   cd <karpenter-core> && KS_FIXTURE=/tmp/fixture.json go test -tags ks_harness -run TestSolveDump ./pkg/controllers/provisioning/scheduling/ > /tmp/go.lines
//   python tools/go_harness/export_fixture.py config3 --pods 20000 --want | diff - <(grep -E '^(NODE|UNSCHEDULED)' /tmp/go.lines)
//
// The scheduler is built the way the reference's own benchmark builds it (scheduling_benchmark_test.go:113-133) -- with a live Topology instead of its inert one, since
// configs[2] is about topology.  Where the two outputs differ only in WHICH equal-count node a pod went to, the difference is Go's unstable sort.Slice(newNodes)
// (scheduler.go:183) and map iteration order: this repository fixes one reachable execution (DESIGN.md section 2); KS_ORDER=1 prints the newNodes order after every
// sort to see exactly where the executions part.  Written against karpenter-core as vendored under /root/reference at the time of writing; untested (no toolchain).
package scheduling_test

import (
	"context"
	"encoding/json"
	"fmt"
	"os"
	"sort"
	"strings"
	"testing"

	"github.com/samber/lo"
	"go.uber.org/zap"
	v1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
	"k8s.io/apimachinery/pkg/types"
	"k8s.io/apimachinery/pkg/util/sets"
	"k8s.io/utils/clock"
	"knative.dev/pkg/logging"
	fakeclient "sigs.k8s.io/controller-runtime/pkg/client/fake"

	"github.com/aws/karpenter-core/pkg/apis/config/settings"
	"github.com/aws/karpenter-core/pkg/apis/v1alpha5"
	"github.com/aws/karpenter-core/pkg/cloudprovider"
	"github.com/aws/karpenter-core/pkg/cloudprovider/fake"
	"github.com/aws/karpenter-core/pkg/controllers/provisioning/scheduling"
	"github.com/aws/karpenter-core/pkg/controllers/state"
	pscheduling "github.com/aws/karpenter-core/pkg/scheduling"
	"github.com/aws/karpenter-core/pkg/test"
)

type fxExpr struct {
	Key    string   `json:"key"`
	Op     string   `json:"op"`
	Values []string `json:"values"`
}
type fxSelector struct {
	MatchLabels      map[string]string `json:"match_labels"`
	MatchExpressions []fxExpr          `json:"match_expressions"`
}
type fxTerm struct {
	TopologyKey   string      `json:"topology_key"`
	LabelSelector *fxSelector `json:"label_selector"`
	Namespaces    []string    `json:"namespaces"`
}
type fxSpread struct {
	MaxSkew           int32       `json:"max_skew"`
	TopologyKey       string      `json:"topology_key"`
	WhenUnsatisfiable string      `json:"when_unsatisfiable"`
	LabelSelector     *fxSelector `json:"label_selector"`
}
type fxToleration struct{ Key, Operator, Value, Effect string }
type fxTaint struct{ Key, Value, Effect string }
type fxContainer struct {
	Requests map[string]string `json:"requests"`
	Limits   map[string]string `json:"limits"`
}
type fxPod struct {
	UID              string            `json:"uid"`
	Namespace        string            `json:"namespace"`
	CreationTs       int64             `json:"creation_ts"`
	Labels           map[string]string `json:"labels"`
	NodeSelector     map[string]string `json:"node_selector"`
	RequiredAffinity [][]fxExpr        `json:"required_affinity"`
	Tolerations      []fxToleration    `json:"tolerations"`
	Containers       []fxContainer     `json:"containers"`
	Spread           []fxSpread        `json:"spread"`
	AffinityRequired []fxTerm          `json:"affinity_required"`
	AntiRequired     []fxTerm          `json:"anti_required"`
}
type fxOffering struct {
	CapacityType string  `json:"capacity_type"`
	Zone         string  `json:"zone"`
	Price        float64 `json:"price"`
	Available    bool    `json:"available"`
}
type fxInstanceType struct {
	Name         string            `json:"name"`
	Requirements []fxExpr          `json:"requirements"`
	Offerings    []fxOffering      `json:"offerings"`
	Capacity     map[string]string `json:"capacity"`
	Overhead     map[string]string `json:"overhead"`
}
type fxProvisioner struct {
	Name          string            `json:"name"`
	Weight        int32             `json:"weight"`
	Labels        map[string]string `json:"labels"`
	Requirements  []fxExpr          `json:"requirements"`
	Taints        []fxTaint         `json:"taints"`
	Limits        map[string]string `json:"limits"`
	InstanceTypes []int             `json:"instance_types"`
}
type fixture struct {
	InstanceTypes []fxInstanceType `json:"instance_types"`
	Provisioners  []fxProvisioner  `json:"provisioners"`
	Pods          []fxPod          `json:"pods"`
}

func rl(m map[string]string) v1.ResourceList {
	out := v1.ResourceList{}
	for k, v := range m {
		out[v1.ResourceName(k)] = resource.MustParse(v)
	}
	return out
}

func sel(s *fxSelector) *metav1.LabelSelector {
	if s == nil {
		return nil
	}
	return &metav1.LabelSelector{MatchLabels: s.MatchLabels, MatchExpressions: lo.Map(s.MatchExpressions, func(e fxExpr, _ int) metav1.LabelSelectorRequirement {
		return metav1.LabelSelectorRequirement{Key: e.Key, Operator: metav1.LabelSelectorOperator(e.Op), Values: e.Values}
	})}
}

func terms(ts []fxTerm) []v1.PodAffinityTerm {
	return lo.Map(ts, func(t fxTerm, _ int) v1.PodAffinityTerm {
		return v1.PodAffinityTerm{TopologyKey: t.TopologyKey, LabelSelector: sel(t.LabelSelector), Namespaces: t.Namespaces}
	})
}

func nsr(es []fxExpr) []v1.NodeSelectorRequirement {
	return lo.Map(es, func(e fxExpr, _ int) v1.NodeSelectorRequirement {
		return v1.NodeSelectorRequirement{Key: e.Key, Operator: v1.NodeSelectorOperator(e.Op), Values: e.Values}
	})
}

func buildPod(p fxPod) *v1.Pod {
	pod := &v1.Pod{ObjectMeta: metav1.ObjectMeta{Name: p.UID, Namespace: p.Namespace, UID: types.UID(p.UID), Labels: p.Labels,
		CreationTimestamp: metav1.Unix(p.CreationTs, 0)}}
	pod.Spec.NodeSelector = p.NodeSelector
	for _, c := range p.Containers {
		pod.Spec.Containers = append(pod.Spec.Containers, v1.Container{Resources: v1.ResourceRequirements{Requests: rl(c.Requests), Limits: rl(c.Limits)}})
	}
	for _, t := range p.Tolerations {
		pod.Spec.Tolerations = append(pod.Spec.Tolerations, v1.Toleration{Key: t.Key, Operator: v1.TolerationOperator(t.Operator), Value: t.Value, Effect: v1.TaintEffect(t.Effect)})
	}
	for _, s := range p.Spread {
		pod.Spec.TopologySpreadConstraints = append(pod.Spec.TopologySpreadConstraints, v1.TopologySpreadConstraint{MaxSkew: s.MaxSkew, TopologyKey: s.TopologyKey,
			WhenUnsatisfiable: v1.UnsatisfiableConstraintAction(s.WhenUnsatisfiable), LabelSelector: sel(s.LabelSelector)})
	}
	if len(p.RequiredAffinity)+len(p.AffinityRequired)+len(p.AntiRequired) > 0 {
		pod.Spec.Affinity = &v1.Affinity{}
		if len(p.RequiredAffinity) > 0 {
			pod.Spec.Affinity.NodeAffinity = &v1.NodeAffinity{RequiredDuringSchedulingIgnoredDuringExecution: &v1.NodeSelector{
				NodeSelectorTerms: lo.Map(p.RequiredAffinity, func(es []fxExpr, _ int) v1.NodeSelectorTerm { return v1.NodeSelectorTerm{MatchExpressions: nsr(es)} })}}
		}
		if len(p.AffinityRequired) > 0 {
			pod.Spec.Affinity.PodAffinity = &v1.PodAffinity{RequiredDuringSchedulingIgnoredDuringExecution: terms(p.AffinityRequired)}
		}
		if len(p.AntiRequired) > 0 {
			pod.Spec.Affinity.PodAntiAffinity = &v1.PodAntiAffinity{RequiredDuringSchedulingIgnoredDuringExecution: terms(p.AntiRequired)}
		}
	}
	return pod
}

func buildInstanceType(t fxInstanceType) *cloudprovider.InstanceType {
	reqs := pscheduling.NewRequirements()
	for _, e := range t.Requirements {
		reqs.Add(pscheduling.NewRequirement(e.Key, v1.NodeSelectorOperator(e.Op), e.Values...))
	}
	return &cloudprovider.InstanceType{Name: t.Name, Requirements: reqs, Capacity: rl(t.Capacity),
		Offerings: lo.Map(t.Offerings, func(o fxOffering, _ int) cloudprovider.Offering {
			return cloudprovider.Offering{CapacityType: o.CapacityType, Zone: o.Zone, Price: o.Price, Available: o.Available}
		}),
		Overhead: &cloudprovider.InstanceTypeOverhead{KubeReserved: rl(t.Overhead)}}
}

func TestSolveDump(t *testing.T) {
	raw, err := os.ReadFile(os.Getenv("KS_FIXTURE"))
	if err != nil {
		t.Skip("KS_FIXTURE not set")
	}
	var fx fixture
	if err := json.Unmarshal(raw, &fx); err != nil {
		t.Fatal(err)
	}
	ctx := logging.WithLogger(context.Background(), zap.NewNop().Sugar())
	ctx = settings.ToContext(ctx, test.Settings())
	all := lo.Map(fx.InstanceTypes, func(it fxInstanceType, _ int) *cloudprovider.InstanceType { return buildInstanceType(it) })
	var provisioners []v1alpha5.Provisioner
	var templates []*scheduling.MachineTemplate
	byProvisioner := map[string][]*cloudprovider.InstanceType{}
	for _, p := range fx.Provisioners {
		opts := test.ProvisionerOptions{ObjectMeta: metav1.ObjectMeta{Name: p.Name}, Labels: p.Labels, Requirements: nsr(p.Requirements),
			Taints: lo.Map(p.Taints, func(x fxTaint, _ int) v1.Taint { return v1.Taint{Key: x.Key, Value: x.Value, Effect: v1.TaintEffect(x.Effect)} })}
		if p.Limits != nil {
			opts.Limits = rl(p.Limits)
		}
		prov := test.Provisioner(opts)
		prov.Spec.Weight = lo.ToPtr(p.Weight)
		provisioners = append(provisioners, *prov)
		byProvisioner[p.Name] = lo.Map(p.InstanceTypes, func(i int, _ int) *cloudprovider.InstanceType { return all[i] })
	}
	// provisioner.go:244-246: templates in weight order (the order this repository's oracle uses: stable by descending weight)
	sort.SliceStable(provisioners, func(i, j int) bool { return lo.FromPtr(provisioners[i].Spec.Weight) > lo.FromPtr(provisioners[j].Spec.Weight) })
	for i := range provisioners {
		templates = append(templates, scheduling.NewMachineTemplate(&provisioners[i]))
	}
	pods := lo.Map(fx.Pods, func(p fxPod, _ int) *v1.Pod { return buildPod(p) })
	cp := fake.NewCloudProvider()
	cp.InstanceTypes = all
	kube := fakeclient.NewClientBuilder().Build() // (NewTopology's countDomains lists the cluster's pods through the client: an empty cluster here, like a fresh provisioning pass)
	cluster := state.NewCluster(&clock.RealClock{}, kube, cp)
	// provisioner.go:267-276: the domain universe from the instance types' requirements and the provisioners' own In requirements
	domains := map[string]sets.String{}
	for _, prov := range provisioners {
		for _, it := range byProvisioner[prov.Name] {
			for key, req := range it.Requirements {
				domains[key] = domains[key].Union(sets.NewString(req.Values()...))
			}
		}
		for key, req := range pscheduling.NewNodeSelectorRequirements(prov.Spec.Requirements...) {
			if req.Operator() == v1.NodeSelectorOpIn {
				domains[key] = domains[key].Union(sets.NewString(req.Values()...))
			}
		}
	}
	topology, err := scheduling.NewTopology(ctx, kube, cluster, domains, pods)
	if err != nil {
		t.Fatal(err)
	}
	s := scheduling.NewScheduler(ctx, kube, templates, provisioners, cluster, nil, topology, byProvisioner, nil, test.NewEventRecorder(), scheduling.SchedulerOptions{})
	nodes, _, err := s.Solve(ctx, pods)
	if err != nil {
		t.Fatal(err)
	}
	placed := map[types.UID]bool{}
	for _, n := range nodes {
		reqs := []string{}
		for k, q := range n.Requests {
			reqs = append(reqs, fmt.Sprintf("%s=%d", k, q.MilliValue()))
		}
		sort.Strings(reqs)
		names := lo.Map(n.InstanceTypeOptions, func(it *cloudprovider.InstanceType, _ int) string { return it.Name })
		sort.Strings(names)
		fmt.Printf("NODE %s | %s | %s | %s\n", n.ProvisionerName, strings.Join(lo.Map(n.Pods, func(p *v1.Pod, _ int) string { placed[p.UID] = true; return string(p.UID) }), " "),
			strings.Join(names, " "), strings.Join(reqs, " "))
	}
	fmt.Printf("UNSCHEDULED %s\n", strings.Join(lo.FilterMap(pods, func(p *v1.Pod, _ int) (string, bool) { return string(p.UID), !placed[p.UID] }), " "))
}
