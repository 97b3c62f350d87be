//go:build knoxgpu

// Drop into internal/operator/filter/ of blockwatch-cc/knoxdb.  The Matcher-level boundary: gpuNumMatcher[T] is a
// filter.Matcher (match.go:15-72) whose MatchVector runs on the MI355X; every other method (WithValue, MatchValue,
// MatchRange, MatchFilter, MatchRangeVectors, Weight, Len, Value ...) is the embedded CPU matcher's, unchanged.
// gpuNumMatcherFactory[T] takes the place of NumMatcherFactory[T] (match_num.go:257-310) in newFactory
// (match.go:74-130) when knoxgpu.Enabled().
// Not compiled in the knoxdb_amd repository (no Go toolchain there); see integration/check_go_calls.py.
package filter

import (
	"blockwatch.cc/knoxdb/internal/bitset"
	"blockwatch.cc/knoxdb/internal/block"
	"blockwatch.cc/knoxdb/internal/knoxgpu"
	"blockwatch.cc/knoxdb/internal/types"
)

var gpuCtx *knoxgpu.Ctx

func init() {
	gpuCtx = knoxgpu.Default() // nil: KNOX_NOGPU or no device -- every factory then returns the CPU matcher
}

type gpuNumMatcherFactory[T Number] struct {
	cpu NumMatcherFactory[T]
}

func (f gpuNumMatcherFactory[T]) New(m FilterMode) Matcher {
	inner := f.cpu.New(m)
	if gpuCtx == nil {
		return inner
	}
	switch m {
	case FilterModeEqual, FilterModeNotEqual, FilterModeGt, FilterModeGe, FilterModeLt, FilterModeLe, FilterModeRange:
		return &gpuNumMatcher[T]{Matcher: inner, typ: f.cpu.typ, mode: m}
	default:
		return inner // IN / NOT IN run on the device through the tree boundary (PhysicalGpuFilter), not per vector
	}
}

// gpuNumMatcher forwards everything to the CPU matcher except MatchVector over a materialized block.
type gpuNumMatcher[T Number] struct {
	Matcher
	typ  types.BlockType
	mode FilterMode
}

func (m *gpuNumMatcher[T]) MatchVector(b *block.Block, bits, mask *bitset.Bitset) {
	// compressed blocks keep the reference's container matchers (match_num.go:351-353); a mask is an optimisation
	// hint the numeric vector kernels ignore (match_num.go:348-350), so it is not forwarded either
	if !b.IsMaterialized() {
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	var a, z T
	switch v := m.Matcher.Value().(type) {
	case RangeValue:
		a, z = v[0].(T), v[1].(T)
	default:
		a = v.(T)
	}
	n, ok := knoxgpu.Cmp(gpuCtx, int(m.typ), int(m.mode), block.NewAccessor[T](b).Slice(), a, z, bits.Bytes())
	if !ok {
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	bits.ResetCount(int(n))
}
