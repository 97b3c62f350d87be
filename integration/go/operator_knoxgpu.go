//go:build knoxgpu

// Drop into internal/operator/ of blockwatch-cc/knoxdb.  The tree-level boundary:
//
//   PhysicalGpuFilter   a PushOperator (operator.go:37-42) with the contract of PhysicalFilter (filter.go:16-58):
//                       Process evaluates the whole filter.Node tree over one package in ONE knox_scan (instead of
//                       filter.Match's per-leaf MatchVector + Bitset.And / Or, match_core.go:14-190) and sets the
//                       package's selection the same way (nil selection when every row matches).
//   GpuAggregateResult  an engine.QueryResultConsumer (engine/interface.go:154-157) with the role of CountResult
//                       (query/result.go:36-60): Append folds the reducers over the selected rows of every package
//                       it is handed; states merge exactly (knox_agg_combine), Value() finalises.
//
// Columns of a package are uploaded once and kept on the device keyed by (pack key, version, field): the second query
// over the same pack reads HBM only.  Encoded (non-materialized) blocks cross PCIe in their container form and are
// decoded on the device (knox_col_upload_raw_containers); []byte / string blocks held as a plain StringPool go up as one
// pack of a bytes column.  Leaves: the compare modes and RANGE on numeric and bytes columns, IN / NOT IN sets, REGEXP on
// bytes columns.  Trees with anything else (wide integers, encoded string containers, empty sets) fall back to the CPU
// operator.  This operator works package by package, as the PushOperator interface hands them out; the
// run-level reader (many packs resident, ONE scan, per-pack results served) is run_knoxgpu.go.
// Not compiled in the knoxdb_amd repository (no Go toolchain there); see integration/check_go_calls.py.
package operator

import (
	"context"

	"blockwatch.cc/knoxdb/internal/bitset"
	"blockwatch.cc/knoxdb/internal/block"
	"blockwatch.cc/knoxdb/internal/encode"
	"blockwatch.cc/knoxdb/internal/knoxgpu"
	"blockwatch.cc/knoxdb/internal/operator/filter"
	"blockwatch.cc/knoxdb/internal/pack"
	"blockwatch.cc/knoxdb/internal/types"
	"blockwatch.cc/knoxdb/pkg/stringx"
)

var (
	_ PushOperator = (*PhysicalGpuFilter)(nil)
)

type colKey struct {
	pack, version uint32
	field         int
}

// deviceCols is the device-resident pack cache of one operator: bounded by the context's arena budget
// (knoxgpu.Ctx.SetArena).  A failing upload evicts the oldest columns of OTHER packages and retries once: columns of the
// package being planned are pinned for the duration of the plan (an earlier leaf's handle is already in the scan's
// column list -- freeing it would hand a dangling knox_col to C); when nothing evictable is left the caller falls back
// to the CPU operator.
type deviceCols struct {
	gpu   *knoxgpu.Ctx
	cols  map[colKey]*knoxgpu.Column
	order []colKey
}

func (d *deviceCols) get(pkg *pack.Package, field int) (*knoxgpu.Column, bool) {
	k := colKey{pkg.Key(), pkg.Version(), field}
	if c, ok := d.cols[k]; ok {
		return c, true
	}
	b := pkg.Block(field)
	if b == nil {
		return nil, false
	}
	c, err := uploadBlock(d.gpu, b)
	if err != nil {
		if d.evictOthers(k.pack, k.version, (len(d.order)+1)/2) == 0 {
			return nil, false // everything resident belongs to this package: nothing to give up
		}
		if c, err = uploadBlock(d.gpu, b); err != nil {
			return nil, false
		}
	}
	d.cols[k] = c
	d.order = append(d.order, k)
	return c, true
}

// evictOthers frees up to n of the oldest columns that do NOT belong to (pack, version); returns how many went.
func (d *deviceCols) evictOthers(pack, version uint32, n int) int {
	kept := d.order[:0]
	freed := 0
	for _, k := range d.order {
		if freed < n && (k.pack != pack || k.version != version) {
			d.cols[k].Free()
			delete(d.cols, k)
			freed++
			continue
		}
		kept = append(kept, k)
	}
	d.order = kept
	return freed
}

func (d *deviceCols) evictAll() {
	for _, k := range d.order {
		d.cols[k].Free()
		delete(d.cols, k)
	}
	d.order = d.order[:0]
}

// containerBytes re-serialises the encoded container of a non-materialized block (NumberContainer.Store,
// internal/encode/container.go:84): the wire form knox_col_upload_raw_containers parses -- the device decodes it.
func containerBytes[T knoxgpu.Number](b *block.Block) ([]byte, bool) {
	c, ok := b.Container().(encode.NumberContainer[T])
	if !ok {
		return nil, false
	}
	return c.Store(make([]byte, 0, c.Size())), true
}

func uploadVector[T knoxgpu.Number](g *knoxgpu.Ctx, b *block.Block) (*knoxgpu.Column, error) {
	if b.IsMaterialized() {
		return knoxgpu.Upload(g, int(b.Type()), block.NewAccessor[T](b).Slice(), 0)
	}
	buf, ok := containerBytes[T](b)
	if !ok {
		return nil, ErrTodo
	}
	return knoxgpu.UploadRawContainers(g, int(b.Type()), [][]byte{buf}, 0)
}

func uploadBlock(g *knoxgpu.Ctx, b *block.Block) (*knoxgpu.Column, error) {
	switch b.Type() {
	case types.BlockInt64:
		return uploadVector[int64](g, b)
	case types.BlockInt32:
		return uploadVector[int32](g, b)
	case types.BlockInt16:
		return uploadVector[int16](g, b)
	case types.BlockInt8:
		return uploadVector[int8](g, b)
	case types.BlockUint64:
		return uploadVector[uint64](g, b)
	case types.BlockUint32:
		return uploadVector[uint32](g, b)
	case types.BlockUint16:
		return uploadVector[uint16](g, b)
	case types.BlockUint8:
		return uploadVector[uint8](g, b)
	case types.BlockFloat64:
		return uploadVector[float64](g, b)
	case types.BlockFloat32:
		return uploadVector[float32](g, b)
	case types.BlockBytes:
		// a plain StringPool goes up as it is (pointer array + buffer = one pack of a bytes column); the dictionary / fixed /
		// compact string containers keep the reference's own matchers (the whole tree then runs on the CPU operator)
		// Container() = the block's vector as `any` (block.go:188): no panic for encoded string containers
		if pool, ok := b.Container().(*stringx.StringPool); ok {
			ptr, buf := pool.Raw() // pkg/stringx accessor quoted in bytes_knoxgpu.go
			return knoxgpu.UploadPool(g, ptr, buf)
		}
	}
	return nil, ErrTodo
}

// plan flattens a filter.Node tree (node.go:29-37) into the postfix form knox_scan takes: operands first, then one
// AND / OR node popping nchild results.  cols receives the device columns in leaf order.  AnyMatch / Skip nodes are
// dropped exactly where MatchAnd / MatchOr skip them (match_core.go:55-58, 129-135).
func (op *PhysicalGpuFilter) plan(n *filter.Node, pkg *pack.Package, nodes []knoxgpu.Pred, cols []*knoxgpu.Column) ([]knoxgpu.Pred, []*knoxgpu.Column, bool) {
	if n.IsLeaf() {
		f := n.Filter
		col, ok := op.dev.get(pkg, f.Index)
		if !ok {
			return nodes, cols, false
		}
		p := knoxgpu.Pred{Kind: knoxgpu.NodeLeaf, Col: len(cols), Op: int(f.Mode)}
		isBytes := f.Type == types.BlockBytes
		switch f.Mode {
		case types.FilterModeEqual, types.FilterModeNotEqual, types.FilterModeGt, types.FilterModeGe,
			types.FilterModeLt, types.FilterModeLe:
			if isBytes {
				if p.Val, ok = f.Value.([]byte); !ok {
					return nodes, cols, false
				}
			} else if p.A, ok = knoxgpu.ValueBits(f.Value); !ok {
				return nodes, cols, false
			}
		case types.FilterModeRange:
			rg, isRange := f.Value.(filter.RangeValue)
			if !isRange {
				return nodes, cols, false
			}
			var okA, okB bool
			if isBytes {
				p.Val, okA = rg[0].([]byte)
				p.Val2, okB = rg[1].([]byte)
			} else {
				p.A, okA = knoxgpu.ValueBits(rg[0])
				p.B, okB = knoxgpu.ValueBits(rg[1])
			}
			if !okA || !okB {
				return nodes, cols, false
			}
		case types.FilterModeIn, types.FilterModeNotIn:
			// the matcher's own view of the set (Value(): a []T / [][]byte whatever WithValue / WithSet was given); an empty
			// set keeps the reference's short cuts on the CPU operator
			if p.Val, ok = knoxgpu.SetBytes(f.Matcher.Value()); !ok || len(p.Val) == 0 {
				return nodes, cols, false
			}
		case types.FilterModeRegexp:
			// bytesRegexpMatcher.Value() = re.String() (match_bytes.go:549-554); a pattern whose result could depend on the
			// Unicode tables of this binary stays on the CPU operator unless the versions agree (knoxgpu.RegexpOnDevice); a
			// pattern outside the device's subset comes back from Scan as an error and takes the same way
			pat, isStr := f.Matcher.Value().(string)
			if !isBytes || !isStr || pat == "" || !knoxgpu.RegexpOnDevice(pat) {
				return nodes, cols, false
			}
			p.Val = []byte(pat)
		default:
			return nodes, cols, false
		}
		return append(nodes, p), append(cols, col), true
	}
	kids := 0
	for _, c := range n.Children {
		if c.IsAnyMatch() || c.Skip {
			continue
		}
		var ok bool
		if nodes, cols, ok = op.plan(c, pkg, nodes, cols); !ok {
			return nodes, cols, false
		}
		kids++
	}
	kind := knoxgpu.NodeAnd
	if n.OrKind {
		kind = knoxgpu.NodeOr
	}
	return append(nodes, knoxgpu.Pred{Kind: kind, NChild: kids}), cols, true
}

type PhysicalGpuFilter struct {
	cpu  *PhysicalFilter // the fallback, same node
	node *filter.Node
	gpu  *knoxgpu.Ctx
	dev  *deviceCols
	bits *bitset.Bitset
	err  error
}

// NewPhysicalGpuFilter has NewPhysicalFilter's signature (filter.go:22-27) plus the device context.
func NewPhysicalGpuFilter(gpu *knoxgpu.Ctx, node *filter.Node, sz int) *PhysicalGpuFilter {
	return &PhysicalGpuFilter{
		cpu:  NewPhysicalFilter(node, sz),
		node: node,
		gpu:  gpu,
		dev:  &deviceCols{gpu: gpu, cols: make(map[colKey]*knoxgpu.Column)},
		bits: bitset.New(sz),
	}
}

func (op *PhysicalGpuFilter) Process(ctx context.Context, src *pack.Package) (*pack.Package, Result) {
	nodes, cols, ok := op.plan(op.node, src, nil, nil)
	if !ok || len(cols) == 0 {
		return op.cpu.Process(ctx, src)
	}
	dbits, err := knoxgpu.NewBitsFor(op.gpu, src.Len(), cols[0])
	if err != nil {
		return op.cpu.Process(ctx, src)
	}
	defer dbits.Free()
	res, err := op.gpu.Scan(nodes, cols, nil, 0, nil, dbits, false, 1)
	if err != nil {
		return op.cpu.Process(ctx, src)
	}
	switch {
	case res.Count == int64(src.Len()):
		src.WithSelection(nil) // PhysicalFilter: `if op.bits.All()` (filter.go:31-32)
	default:
		sel, err := dbits.Indexes(make([]uint32, 0, res.Count))
		if err != nil {
			return op.cpu.Process(ctx, src)
		}
		src.WithSelection(sel)
	}
	return src, ResultOK
}

func (op *PhysicalGpuFilter) Finalize(ctx context.Context) error { return nil }
func (op *PhysicalGpuFilter) Err() error                         { return op.err }

func (op *PhysicalGpuFilter) Close() {
	op.dev.evictAll()
	op.cpu.Close()
	op.node = nil
	op.bits.Close()
	op.bits = nil
	op.err = nil
}

// GpuAggregateResult folds reducers over the rows a filter selected, package by package.
type GpuAggregateResult struct {
	gpu    *knoxgpu.Ctx
	node   *filter.Node
	filter *PhysicalGpuFilter
	aggs   []knoxgpu.Agg // Col = the FIELD index here; remapped to the scan's column list per package
	states [][]knoxgpu.State
	rows   int64
	n      int
}

// NewGpuAggregateResult: aggs name a reducer (knoxgpu.RSum ...) and a field index of the package schema.
func NewGpuAggregateResult(gpu *knoxgpu.Ctx, node *filter.Node, aggs []knoxgpu.Agg) *GpuAggregateResult {
	return &GpuAggregateResult{gpu: gpu, node: node, filter: NewPhysicalGpuFilter(gpu, node, 0), aggs: aggs,
		states: make([][]knoxgpu.State, len(aggs))}
}

// Append runs filter + reducers over one package in one fused scan; rows ids are offset by the rows seen so far, so
// first / last / min / max report positions in stream order (rank order = row order across packages).
func (r *GpuAggregateResult) Append(ctx context.Context, src *pack.Package) error {
	nodes, cols, ok := r.filter.plan(r.node, src, nil, nil)
	if !ok {
		return ErrTodo // a leaf kind the device plan does not take (the caller keeps its CPU consumer for such plans)
	}
	aggs := make([]knoxgpu.Agg, len(r.aggs))
	for i, a := range r.aggs {
		c, ok := r.filter.dev.get(src, a.Col) // materialized vector or encoded container: both upload (uploadVector)
		if !ok {
			return ErrTodo
		}
		aggs[i] = knoxgpu.Agg{Func: a.Func, Col: len(cols)}
		cols = append(cols, c)
	}
	res, err := r.gpu.Scan(nodes, cols, aggs, uint64(r.rows), nil, nil, false, 1)
	if err != nil {
		return err
	}
	for i := range aggs {
		r.states[i] = append(r.states[i], res.States[i])
	}
	r.rows += int64(src.Len())
	r.n += int(res.Count)
	return nil
}

func (r *GpuAggregateResult) Len() int { return r.n }

// Value finalises aggregate i: the reducer's Value() bits (of the field's type) and its `ok`.
func (r *GpuAggregateResult) Value(i int) (uint64, bool) {
	if len(r.states[i]) == 0 {
		return 0, false
	}
	st := knoxgpu.Combine(r.states[i])
	bits, seen, _ := st.Value(r.aggs[i].Func)
	return bits, seen
}

func (r *GpuAggregateResult) Close() { r.filter.Close() }
