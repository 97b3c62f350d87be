//go:build knoxgpu

// Drop into internal/cmp/: rebinds the exported compare func vars (cmp.go:6-114) to the GPU
// path, the same way cmp_amd64.go:14-218 rebinds them to AVX2 / AVX-512.  NOT compiled here.
package cmp

import "blockwatch.cc/knoxdb/internal/knoxgpu"

var gpu *knoxgpu.Ctx

const (
	blockInt64, blockUint64, blockFloat64 = 1, 5, 9 // types.BlockType
	modeEq, modeNe, modeGt, modeGe, modeLt, modeLe, modeRange = 1, 2, 3, 4, 5, 6, 9
)

func init() {
	if !knoxgpu.Enabled {
		return
	}
	ctx, err := knoxgpu.Open(0)
	if err != nil {
		return // no device: keep the CPU kernels already installed by cmp_amd64.go
	}
	gpu = ctx
	Int64Equal = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeEq, src, val, 0, bits) }
	Int64NotEqual = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeNe, src, val, 0, bits) }
	Int64Less = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeLt, src, val, 0, bits) }
	Int64LessEqual = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeLe, src, val, 0, bits) }
	Int64Greater = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeGt, src, val, 0, bits) }
	Int64GreaterEqual = func(src []int64, val int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeGe, src, val, 0, bits) }
	Int64Between = func(src []int64, a, b int64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockInt64, modeRange, src, a, b, bits) }
	Float64Less = func(src []float64, val float64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockFloat64, modeLt, src, val, 0, bits) }
	Float64Between = func(src []float64, a, b float64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockFloat64, modeRange, src, a, b, bits) }
	Uint64Equal = func(src []uint64, val uint64, bits []byte) int64 { return knoxgpu.Cmp(gpu, blockUint64, modeEq, src, val, 0, bits) }
	// ... the remaining vars follow the same pattern (70 in total)
}
