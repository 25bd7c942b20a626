"""karpenter_core_amd -- MI355X-native drop-in for karpenter-core's provisioning scheduler hot path
(`scheduling.Scheduler.Solve` + `scheduling.Requirements`).  See DESIGN.md."""
__version__ = "0.1.0"
