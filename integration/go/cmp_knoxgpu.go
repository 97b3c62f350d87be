//go:build knoxgpu

// Drop into internal/cmp/ of blockwatch-cc/knoxdb.  Rebinds ALL 84 exported compare func vars (cmp.go:6-114) to the
// MI355X path, the same way cmp_amd64.go:14-218 rebinds them to AVX2 / AVX-512: 10 numeric types x 7 modes go through
// knox_cmp (host slices in, host bitset out), Int128 / Int256 x 7 modes through knox_match_wide over the stride's limb
// slices.  A failing GPU call falls back to the kernel that was installed before (the generic / SIMD one).
// Not compiled in the knoxdb_amd repository (no Go toolchain there); integration/check_go_calls.py type-checks every
// C call in it against include/knoxgpu.h, tests/cpp/test_host.cpp drives the same C entry points on the GPU.
package cmp

import (
	"blockwatch.cc/knoxdb/internal/knoxgpu"
	"blockwatch.cc/knoxdb/pkg/num"
)

// the contexts the rebound kernels run on: a small pool, so that concurrent queries (one goroutine each) do not queue
// behind one mutex (knoxgpu.Pool; KNOX_CMP_CONTEXTS, default 4)
var gpus *knoxgpu.Pool

// types.BlockType (internal/types/block.go:20-36) and types.FilterMode (internal/types/mode.go:13-27)
const (
	blockInt64, blockInt32, blockInt16, blockInt8       = 1, 2, 3, 4
	blockUint64, blockUint32, blockUint16, blockUint8   = 5, 6, 7, 8
	blockFloat64, blockFloat32                          = 9, 10
	modeEq, modeNe, modeGt, modeGe, modeLt, modeLe      = 1, 2, 3, 4, 5, 6
	modeRange                                           = 9
)

// one value / range kernel with its fallback.  Below knoxgpu.CmpMinRows rows the call never leaves the CPU: the kernel
// installed before (cmp_amd64.go's AVX2 / AVX-512 one) answers a production pack in a few microseconds, a device round
// trip cannot (measured table: DESIGN.md 7, tools/cmp_latency.py).
func bind1[T knoxgpu.Number](dtype, mode int, prev func([]T, T, []byte) int64) func([]T, T, []byte) int64 {
	return func(src []T, val T, bits []byte) int64 {
		if len(src) < knoxgpu.CmpMinRows {
			return prev(src, val, bits)
		}
		gpu := gpus.Get()
		defer gpus.Put(gpu)
		if n, ok := knoxgpu.Cmp(gpu, dtype, mode, src, val, val, bits); ok {
			return n
		}
		return prev(src, val, bits)
	}
}

func bind2[T knoxgpu.Number](dtype int, prev func([]T, T, T, []byte) int64) func([]T, T, T, []byte) int64 {
	return func(src []T, a, b T, bits []byte) int64 {
		if len(src) < knoxgpu.CmpMinRows {
			return prev(src, a, b, bits)
		}
		gpu := gpus.Get()
		defer gpus.Put(gpu)
		if n, ok := knoxgpu.Cmp(gpu, dtype, modeRange, src, a, b, bits); ok {
			return n
		}
		return prev(src, a, b, bits)
	}
}

func bind128(mode int, prev func(*num.Int128Stride, num.Int128, []byte, []byte) int64) func(*num.Int128Stride, num.Int128, []byte, []byte) int64 {
	return func(src *num.Int128Stride, val num.Int128, bits, mask []byte) int64 {
		if src.Len() < knoxgpu.CmpMinRows {
			return prev(src, val, bits, mask)
		}
		gpu := gpus.Get()
		defer gpus.Put(gpu)
		if n, ok := knoxgpu.MatchWide(gpu, [][]uint64{knoxgpu.AsU64(src.X0), src.X1}, mode, val[:], nil, mask, bits, src.Len()); ok {
			return n
		}
		return prev(src, val, bits, mask)
	}
}

func bind256(mode int, prev func(*num.Int256Stride, num.Int256, []byte, []byte) int64) func(*num.Int256Stride, num.Int256, []byte, []byte) int64 {
	return func(src *num.Int256Stride, val num.Int256, bits, mask []byte) int64 {
		if src.Len() < knoxgpu.CmpMinRows {
			return prev(src, val, bits, mask)
		}
		gpu := gpus.Get()
		defer gpus.Put(gpu)
		if n, ok := knoxgpu.MatchWide(gpu, [][]uint64{knoxgpu.AsU64(src.X0), src.X1, src.X2, src.X3}, mode, val[:], nil, mask, bits, src.Len()); ok {
			return n
		}
		return prev(src, val, bits, mask)
	}
}

func init() {
	if !knoxgpu.Enabled() { // KNOX_NOGPU=1: keep the CPU kernels (the switch of internal/cpu/avx.go:43-89)
		return
	}
	pool, err := knoxgpu.NewPool(0, knoxgpu.PoolSize())
	if err != nil {
		return // no device: keep the CPU kernels already installed by cmp_amd64.go
	}
	gpus = pool

	// uint8
	Uint8Equal = bind1[uint8](blockUint8, modeEq, Uint8Equal)
	Uint8NotEqual = bind1[uint8](blockUint8, modeNe, Uint8NotEqual)
	Uint8Less = bind1[uint8](blockUint8, modeLt, Uint8Less)
	Uint8LessEqual = bind1[uint8](blockUint8, modeLe, Uint8LessEqual)
	Uint8Greater = bind1[uint8](blockUint8, modeGt, Uint8Greater)
	Uint8GreaterEqual = bind1[uint8](blockUint8, modeGe, Uint8GreaterEqual)
	Uint8Between = bind2[uint8](blockUint8, Uint8Between)
	// uint16
	Uint16Equal = bind1[uint16](blockUint16, modeEq, Uint16Equal)
	Uint16NotEqual = bind1[uint16](blockUint16, modeNe, Uint16NotEqual)
	Uint16Less = bind1[uint16](blockUint16, modeLt, Uint16Less)
	Uint16LessEqual = bind1[uint16](blockUint16, modeLe, Uint16LessEqual)
	Uint16Greater = bind1[uint16](blockUint16, modeGt, Uint16Greater)
	Uint16GreaterEqual = bind1[uint16](blockUint16, modeGe, Uint16GreaterEqual)
	Uint16Between = bind2[uint16](blockUint16, Uint16Between)
	// uint32
	Uint32Equal = bind1[uint32](blockUint32, modeEq, Uint32Equal)
	Uint32NotEqual = bind1[uint32](blockUint32, modeNe, Uint32NotEqual)
	Uint32Less = bind1[uint32](blockUint32, modeLt, Uint32Less)
	Uint32LessEqual = bind1[uint32](blockUint32, modeLe, Uint32LessEqual)
	Uint32Greater = bind1[uint32](blockUint32, modeGt, Uint32Greater)
	Uint32GreaterEqual = bind1[uint32](blockUint32, modeGe, Uint32GreaterEqual)
	Uint32Between = bind2[uint32](blockUint32, Uint32Between)
	// uint64
	Uint64Equal = bind1[uint64](blockUint64, modeEq, Uint64Equal)
	Uint64NotEqual = bind1[uint64](blockUint64, modeNe, Uint64NotEqual)
	Uint64Less = bind1[uint64](blockUint64, modeLt, Uint64Less)
	Uint64LessEqual = bind1[uint64](blockUint64, modeLe, Uint64LessEqual)
	Uint64Greater = bind1[uint64](blockUint64, modeGt, Uint64Greater)
	Uint64GreaterEqual = bind1[uint64](blockUint64, modeGe, Uint64GreaterEqual)
	Uint64Between = bind2[uint64](blockUint64, Uint64Between)
	// int8
	Int8Equal = bind1[int8](blockInt8, modeEq, Int8Equal)
	Int8NotEqual = bind1[int8](blockInt8, modeNe, Int8NotEqual)
	Int8Less = bind1[int8](blockInt8, modeLt, Int8Less)
	Int8LessEqual = bind1[int8](blockInt8, modeLe, Int8LessEqual)
	Int8Greater = bind1[int8](blockInt8, modeGt, Int8Greater)
	Int8GreaterEqual = bind1[int8](blockInt8, modeGe, Int8GreaterEqual)
	Int8Between = bind2[int8](blockInt8, Int8Between)
	// int16
	Int16Equal = bind1[int16](blockInt16, modeEq, Int16Equal)
	Int16NotEqual = bind1[int16](blockInt16, modeNe, Int16NotEqual)
	Int16Less = bind1[int16](blockInt16, modeLt, Int16Less)
	Int16LessEqual = bind1[int16](blockInt16, modeLe, Int16LessEqual)
	Int16Greater = bind1[int16](blockInt16, modeGt, Int16Greater)
	Int16GreaterEqual = bind1[int16](blockInt16, modeGe, Int16GreaterEqual)
	Int16Between = bind2[int16](blockInt16, Int16Between)
	// int32
	Int32Equal = bind1[int32](blockInt32, modeEq, Int32Equal)
	Int32NotEqual = bind1[int32](blockInt32, modeNe, Int32NotEqual)
	Int32Less = bind1[int32](blockInt32, modeLt, Int32Less)
	Int32LessEqual = bind1[int32](blockInt32, modeLe, Int32LessEqual)
	Int32Greater = bind1[int32](blockInt32, modeGt, Int32Greater)
	Int32GreaterEqual = bind1[int32](blockInt32, modeGe, Int32GreaterEqual)
	Int32Between = bind2[int32](blockInt32, Int32Between)
	// int64
	Int64Equal = bind1[int64](blockInt64, modeEq, Int64Equal)
	Int64NotEqual = bind1[int64](blockInt64, modeNe, Int64NotEqual)
	Int64Less = bind1[int64](blockInt64, modeLt, Int64Less)
	Int64LessEqual = bind1[int64](blockInt64, modeLe, Int64LessEqual)
	Int64Greater = bind1[int64](blockInt64, modeGt, Int64Greater)
	Int64GreaterEqual = bind1[int64](blockInt64, modeGe, Int64GreaterEqual)
	Int64Between = bind2[int64](blockInt64, Int64Between)
	// float32
	Float32Equal = bind1[float32](blockFloat32, modeEq, Float32Equal)
	Float32NotEqual = bind1[float32](blockFloat32, modeNe, Float32NotEqual)
	Float32Less = bind1[float32](blockFloat32, modeLt, Float32Less)
	Float32LessEqual = bind1[float32](blockFloat32, modeLe, Float32LessEqual)
	Float32Greater = bind1[float32](blockFloat32, modeGt, Float32Greater)
	Float32GreaterEqual = bind1[float32](blockFloat32, modeGe, Float32GreaterEqual)
	Float32Between = bind2[float32](blockFloat32, Float32Between)
	// float64
	Float64Equal = bind1[float64](blockFloat64, modeEq, Float64Equal)
	Float64NotEqual = bind1[float64](blockFloat64, modeNe, Float64NotEqual)
	Float64Less = bind1[float64](blockFloat64, modeLt, Float64Less)
	Float64LessEqual = bind1[float64](blockFloat64, modeLe, Float64LessEqual)
	Float64Greater = bind1[float64](blockFloat64, modeGt, Float64Greater)
	Float64GreaterEqual = bind1[float64](blockFloat64, modeGe, Float64GreaterEqual)
	Float64Between = bind2[float64](blockFloat64, Float64Between)
	// int128: num.Int128Stride limbs, most significant first (pkg/num/int128_stride.go)
	Int128Equal = bind128(modeEq, Int128Equal)
	Int128NotEqual = bind128(modeNe, Int128NotEqual)
	Int128Less = bind128(modeLt, Int128Less)
	Int128LessEqual = bind128(modeLe, Int128LessEqual)
	Int128Greater = bind128(modeGt, Int128Greater)
	Int128GreaterEqual = bind128(modeGe, Int128GreaterEqual)
	{
		prev := Int128Between
		Int128Between = func(src *num.Int128Stride, a, b num.Int128, bits, mask []byte) int64 {
			if src.Len() < knoxgpu.CmpMinRows {
				return prev(src, a, b, bits, mask)
			}
			gpu := gpus.Get()
			defer gpus.Put(gpu)
			limbs := [][]uint64{knoxgpu.AsU64(src.X0), src.X1}
			if n, ok := knoxgpu.MatchWide(gpu, limbs, modeRange, a[:], b[:], mask, bits, src.Len()); ok {
				return n
			}
			return prev(src, a, b, bits, mask)
		}
	}
	// int256: num.Int256Stride limbs, most significant first (pkg/num/int256_stride.go)
	Int256Equal = bind256(modeEq, Int256Equal)
	Int256NotEqual = bind256(modeNe, Int256NotEqual)
	Int256Less = bind256(modeLt, Int256Less)
	Int256LessEqual = bind256(modeLe, Int256LessEqual)
	Int256Greater = bind256(modeGt, Int256Greater)
	Int256GreaterEqual = bind256(modeGe, Int256GreaterEqual)
	{
		prev := Int256Between
		Int256Between = func(src *num.Int256Stride, a, b num.Int256, bits, mask []byte) int64 {
			if src.Len() < knoxgpu.CmpMinRows {
				return prev(src, a, b, bits, mask)
			}
			gpu := gpus.Get()
			defer gpus.Put(gpu)
			limbs := [][]uint64{knoxgpu.AsU64(src.X0), src.X1, src.X2, src.X3}
			if n, ok := knoxgpu.MatchWide(gpu, limbs, modeRange, a[:], b[:], mask, bits, src.Len()); ok {
				return n
			}
			return prev(src, a, b, bits, mask)
		}
	}
}
