// ks_algebra.h -- the Requirement set algebra on the bitmask encoding (host + device).
//
// One requirement on one key is {present, complement, mask, gt, lt}: `mask` bit i says value i of the
// key's universe is in the reference's `values` set (requirement.go:36-42).  Bounds use sentinels
// (INT32_MIN = no greaterThan, INT32_MAX = no lessThan) so max()/min() implement maxIntPtr/minIntPtr
// (requirement.go:245-269) without branches.  Every function cites the Go it restates; the whole file
// is checked cell-by-cell against the reference truth tables (tests/test_algebra_tables.py), on the
// host and on the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KS_HD __host__ __device__ __forceinline__
#else
#define KS_HD inline
#endif

#define KS_NOGT INT32_MIN
#define KS_NOLT INT32_MAX

struct KReq {
  uint64_t mask;
  int32_t gt, lt;
  bool present, complement;
};

KS_HD KReq kreq_absent() { KReq r; r.mask = 0; r.gt = KS_NOGT; r.lt = KS_NOLT; r.present = false; r.complement = false; return r; }
KS_HD KReq kreq_exists() { KReq r = kreq_absent(); r.present = true; r.complement = true; return r; }   // NewRequirement(key, Exists)
KS_HD KReq kreq_in(uint64_t mask) { KReq r = kreq_absent(); r.present = true; r.mask = mask; return r; } // In [values] / DoesNotExist when 0

// withinIntPtrs over the whole universe (requirement.go:227-243): bit v set iff value v passes the
// bounds.  No bounds -> every value (also non-integers) passes.
KS_HD uint64_t kreq_within_mask(const int32_t* value_int, uint32_t nvalues, int32_t gt, int32_t lt) {
  if (gt == KS_NOGT && lt == KS_NOLT) return ~0ull;
  uint64_t w = 0;
  for (uint32_t v = 0; v < nvalues; ++v) {
    int32_t x = value_int[v];
    if (x == INT32_MIN) continue;                 // not an integer -> invalid once bounds are set
    if (gt != KS_NOGT && gt >= x) continue;
    if (lt != KS_NOLT && lt <= x) continue;
    w |= 1ull << v;
  }
  return w;
}

// Len()==0 (requirement.go:199-204): only a concrete empty set is empty.
KS_HD bool kreq_len0(const KReq& r) { return !r.complement && r.mask == 0; }
// Operator() in {NotIn, DoesNotExist} (requirement.go:186-197): a complement set with excluded
// values, or a concrete empty set.  A complement set with bounds but no excluded values is Exists.
KS_HD bool kreq_nidne(const KReq& r) { return r.complement ? r.mask != 0 : r.mask == 0; }

// Len() (requirement.go:199-204): |values| for a concrete set, MaxInt64 - |excluded values| for a complement set.
KS_HD int64_t kreq_len(const KReq& r) { const int64_t n = (int64_t)__builtin_popcountll(r.mask); return r.complement ? INT64_MAX - n : n; }
// Operator() (requirement.go:186-197) as 0 In, 1 NotIn, 2 Exists, 3 DoesNotExist -- derived from (complement, Len) exactly like the reference;
// kreq_nidne / kreq_len0 above are the two predicates of it the kernels use.
KS_HD int kreq_operator(const KReq& r) { if (r.complement) return kreq_len(r) < INT64_MAX ? 1 : 2; return kreq_len(r) > 0 ? 0 : 3; }

// Intersection (requirement.go:117-150); both operands present.
KS_HD KReq kreq_intersect(const KReq& a, const KReq& b, const int32_t* value_int, uint32_t nvalues) {
  KReq r; r.present = true;
  r.complement = a.complement && b.complement;
  int32_t gt = a.gt > b.gt ? a.gt : b.gt;
  int32_t lt = a.lt < b.lt ? a.lt : b.lt;
  if (gt != KS_NOGT && lt != KS_NOLT && gt >= lt) { r.complement = false; r.mask = 0; r.gt = KS_NOGT; r.lt = KS_NOLT; return r; }
  uint64_t vals;
  if (a.complement && b.complement) vals = a.mask | b.mask;
  else if (a.complement && !b.complement) vals = b.mask & ~a.mask;
  else if (!a.complement && b.complement) vals = a.mask & ~b.mask;
  else vals = a.mask & b.mask;
  vals &= kreq_within_mask(value_int, nvalues, gt, lt);
  r.mask = vals;
  if (r.complement) { r.gt = gt; r.lt = lt; } else { r.gt = KS_NOGT; r.lt = KS_NOLT; }
  return r;
}

// Requirements.Add on one key (requirements.go:87-94): absent side contributes nothing.
KS_HD KReq kreq_add(const KReq& existing, const KReq& incoming, const int32_t* value_int, uint32_t nvalues) {
  if (!incoming.present) return existing;
  if (!existing.present) return incoming;
  return kreq_intersect(incoming, existing, value_int, nvalues);
}

// Has over the universe: bit v iff r.Has(value v) (requirement.go:171-176).
KS_HD uint64_t kreq_has_mask(const KReq& r, const int32_t* value_int, uint32_t nvalues) {
  uint64_t univ = nvalues >= 64 ? ~0ull : ((1ull << nvalues) - 1);
  uint64_t in = r.complement ? ~r.mask : r.mask;
  return in & univ & kreq_within_mask(value_int, nvalues, r.gt, r.lt);
}

// Requirements.Intersects on one key (requirements.go:189-206): true == error.
KS_HD bool kreq_intersects_fail(const KReq& existing, const KReq& incoming, const int32_t* value_int, uint32_t nvalues) {
  if (!existing.present || !incoming.present) return false;
  if (!kreq_len0(kreq_intersect(existing, incoming, value_int, nvalues))) return false;
  return !(kreq_nidne(incoming) && kreq_nidne(existing));
}

// Requirements.Compatible on one key (requirements.go:123-133): true == error.
KS_HD bool kreq_compatible_fail(const KReq& receiver, const KReq& incoming, bool well_known, const int32_t* value_int, uint32_t nvalues) {
  if (!incoming.present) return false;
  if (!well_known && !receiver.present && !kreq_nidne(incoming)) return true;   // "label does not have known values"
  return kreq_intersects_fail(receiver, incoming, value_int, nvalues);
}
