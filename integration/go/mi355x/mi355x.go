// Package mi355x is the Go side of the drop-in: a provider.VectorStore / VectorStoreTable
// (provider/online.go:42-70) backed by the MI355X engine through cgo over include/ehx.h.
//
// SOURCE ONLY: the build image has no Go toolchain (`go version`: not found), so this file is not
// compiled or tested here; it is kept minimal on purpose (one C call per interface method, no
// pointers retained by C after return, errors mapped to the fferr types the conformance suites
// provider/vectorstore_test.go:22-46 and provider/online_test.go:28-70 assert on).
//
// Wiring on the reference side (three lines):
//   provider/provider_type/provider_type.go:16-26   add  MI355XOnline Type = "MI355X_ONLINE"
//   provider/provider.go:23-48                      add  pt.MI355XOnline: mi355xOnlineStoreFactory
//   go build with  CGO_CFLAGS=-I<repo>/include  CGO_LDFLAGS="-L<repo>/embeddinghub_amd/lib -lehx"
package provider

/*
#cgo LDFLAGS: -lehx
#include <stdlib.h>
#include "ehx.h"
*/
import "C"

import (
	"encoding/json"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/featureform/fferr"
	pc "github.com/featureform/provider/provider_config"
	pt "github.com/featureform/provider/provider_type"
	"github.com/featureform/provider/types"
)

// MI355XConfig is the JSON blob carried in metadata (cf. pc.RedisConfig, redis_config.go:18-39).
type MI355XConfig struct {
	Devices []int  `json:"devices"` // HIP device ids (ehx_init): unsharded tables live on Devices[0]
	Shards  uint32 `json:"shards"`  // > 1: every table is row-sharded over Devices inside this process (ehx_params.shards)
	Metric  string `json:"metric"`  // "cosine" (default, as redis.go:253 / pinecone.go:251), "l2", "ip"
	Mode    string `json:"mode"`    // "flat" (exact, default) | "graph"
	EF      uint32 `json:"ef"`
	// graph mode: rows per concurrent insertion round of BatchSet (ehx_params.build_batch; 0 = strictly
	// sequential, the graph single-threaded hnswlib would build; materialization wants e.g. 4096)
	BuildBatch uint32 `json:"build_batch"`
	// graph mode: expansions per search step (ehx_params.search_width; 0 / 1 = hnswlib's order, the default; 2 / 4 = the
	// wide walk, a throughput mode with recall within BASELINE's gate of the strict walk — INTEGRATION.md §5)
	SearchWidth uint32 `json:"search_width"`
}

type mi355xOnlineStore struct {
	cfg MI355XConfig
	BaseProvider
}

// ehx_last_error() is a THREAD-LOCAL string and a goroutine may change OS threads between two cgo calls: every method
// that reports an engine error pins its goroutine for the duration of the failing call and the read of its message.
func pinned() func() {
	runtime.LockOSThread()
	return runtime.UnlockOSThread
}

func mi355xOnlineStoreFactory(serialized pc.SerializedConfig) (Provider, error) {
	var cfg MI355XConfig
	if err := json.Unmarshal(serialized, &cfg); err != nil {
		return nil, fferr.NewInternalError(err)
	}
	var dev *C.int
	devs := make([]C.int, len(cfg.Devices))
	for i, d := range cfg.Devices {
		devs[i] = C.int(d)
	}
	if len(devs) > 0 {
		dev = &devs[0]
	}
	defer pinned()()
	if rc := C.ehx_init(dev, C.int(len(devs))); rc != C.EHX_OK {
		return nil, fferr.NewConnectionError(string(pt.MI355XOnline), fmt.Errorf("%s", C.GoString(C.ehx_last_error())))
	}
	return &mi355xOnlineStore{cfg: cfg, BaseProvider: BaseProvider{ProviderType: pt.MI355XOnline, ProviderConfig: serialized}}, nil
}

func (s *mi355xOnlineStore) AsOnlineStore() (OnlineStore, error) { return s, nil }
func (s *mi355xOnlineStore) Close() error                        { return nil }

// Engine state lives in the process-global registry keyed by the space name, because serving
// builds a fresh provider object per request (serving/serving.go:779-794).
func spaceName(feature, variant string) string { return fmt.Sprintf("Featureform_table__%s__%s", feature, variant) }

func (s *mi355xOnlineStore) metric() C.int {
	switch s.cfg.Metric {
	case "l2":
		return C.EHX_METRIC_L2SQ
	case "ip":
		return C.EHX_METRIC_IP
	default:
		return C.EHX_METRIC_COSINE
	}
}

func (s *mi355xOnlineStore) open(feature, variant string) (*mi355xTable, error) {
	name := spaceName(feature, variant)
	cname := C.CString(name)
	defer C.free(unsafe.Pointer(cname))
	var sp *C.ehx_space
	if rc := C.ehx_space_open(cname, C.size_t(len(name)), &sp); rc != C.EHX_OK {
		return nil, fferr.NewDatasetNotFoundError(feature, variant, nil) // redis.go:103-104
	}
	var dims C.uint32_t
	C.ehx_space_dims(sp, &dims)
	return &mi355xTable{sp: sp, dims: int(dims), feature: feature, variant: variant}, nil
}

func (s *mi355xOnlineStore) create(feature, variant string, dims int32) (*mi355xTable, error) {
	name := spaceName(feature, variant)
	cname := C.CString(name)
	defer C.free(unsafe.Pointer(cname))
	defer pinned()()
	var p C.ehx_params
	p.shards = C.uint32_t(s.cfg.Shards)
	if s.cfg.Mode == "graph" {
		p.mode = C.EHX_MODE_GRAPH
	}
	p.ef = C.uint32_t(s.cfg.EF)
	p.build_batch = C.uint32_t(s.cfg.BuildBatch)
	p.search_width = C.uint32_t(s.cfg.SearchWidth)
	var sp *C.ehx_space
	rc := C.ehx_space_create(cname, C.size_t(len(name)), C.uint32_t(dims), s.metric(), C.EHX_DTYPE_F32, &p, &sp)
	switch rc {
	case C.EHX_OK:
		return &mi355xTable{sp: sp, dims: int(dims), feature: feature, variant: variant}, nil
	case C.EHX_EEXISTS:
		return nil, fferr.NewDatasetAlreadyExistsError(feature, variant, nil) // online_test.go:91-103
	default:
		return nil, fferr.NewResourceExecutionError(string(pt.MI355XOnline), feature, variant, fferr.FEATURE_VARIANT,
			fmt.Errorf("%s", C.GoString(C.ehx_last_error())))
	}
}

// VectorStore (online.go:55-59)
func (s *mi355xOnlineStore) CreateIndex(feature, variant string, vectorType types.VectorType) (VectorStoreTable, error) {
	if t, err := s.open(feature, variant); err == nil { // CreateIndex then CreateTable (materialize.go:128-152)
		return t, nil
	}
	return s.create(feature, variant, vectorType.Dimension)
}
func (s *mi355xOnlineStore) DeleteIndex(feature, variant string) error { return s.DeleteTable(feature, variant) }

// OnlineStore (online.go:42-48)
func (s *mi355xOnlineStore) GetTable(feature, variant string) (OnlineStoreTable, error) {
	return s.open(feature, variant)
}
func (s *mi355xOnlineStore) CreateTable(feature, variant string, valueType types.ValueType) (OnlineStoreTable, error) {
	vt, ok := valueType.(types.VectorType)
	if !ok {
		return nil, fferr.NewDataTypeNotFoundErrorf(valueType, "the MI355X store holds embedding vectors only")
	}
	if t, err := s.open(feature, variant); err == nil {
		n := C.uint64_t(0)
		C.ehx_space_size(t.sp, &n)
		if n > 0 || !vt.IsEmbedding {
			return nil, fferr.NewDatasetAlreadyExistsError(feature, variant, nil)
		}
		return t, nil // index created by CreateIndex a moment ago
	}
	return s.create(feature, variant, vt.Dimension)
}
func (s *mi355xOnlineStore) DeleteTable(feature, variant string) error {
	t, err := s.open(feature, variant)
	if err != nil {
		return err
	}
	C.ehx_space_drop(t.sp)
	return nil
}

type mi355xTable struct {
	sp               *C.ehx_space
	dims             int
	feature, variant string
}

func (t *mi355xTable) fail() error {
	return fferr.NewResourceExecutionError(string(pt.MI355XOnline), t.feature, t.variant, fferr.FEATURE_VARIANT,
		fmt.Errorf("%s", C.GoString(C.ehx_last_error())))
}

// OnlineStoreTable.Set (online.go:51): called from 500 goroutines per chunk (runner/copy.go:34); the engine combines
// concurrent single-row Sets into batches (write-combiner in ehx_set; integration/c/set_hammer.c is the C mirror of
// this call pattern and runs in the GPU test suite).
func (t *mi355xTable) Set(entity string, value interface{}) error {
	vec, ok := value.([]float32)
	if !ok || len(vec) != t.dims {
		return fferr.NewDataTypeNotFoundErrorf(value, "expected []float32 of length %d", t.dims) // redis.go:408-413
	}
	defer pinned()()
	ckey := C.CString(entity)
	defer C.free(unsafe.Pointer(ckey))
	if rc := C.ehx_set(t.sp, ckey, C.size_t(len(entity)), (*C.float)(unsafe.Pointer(&vec[0]))); rc != C.EHX_OK {
		return t.fail()
	}
	return nil
}

// BatchOnlineTable (online.go:66-70): lets runner/copy.go:99-144 take its batch branch.
func (t *mi355xTable) MaxBatchSize() (int, error) { return 65536, nil }
func (t *mi355xTable) BatchSet(items []SetItem) error {
	n := len(items)
	if n == 0 {
		return nil
	}
	defer pinned()()
	flat := make([]float32, 0, n*t.dims)
	keys := make([]*C.char, n)
	lens := make([]C.size_t, n)
	for i, it := range items {
		vec, ok := it.Value.([]float32)
		if !ok || len(vec) != t.dims {
			return fferr.NewDataTypeNotFoundErrorf(it.Value, "expected []float32 of length %d", t.dims)
		}
		flat = append(flat, vec...)
		keys[i] = C.CString(it.Entity)
		lens[i] = C.size_t(len(it.Entity))
	}
	defer func() {
		for _, k := range keys {
			C.free(unsafe.Pointer(k))
		}
	}()
	// keys is a Go slice of C pointers: allowed by the cgo pointer rules (no Go pointers inside)
	if rc := C.ehx_set_batch(t.sp, C.size_t(n), (**C.char)(unsafe.Pointer(&keys[0])), &lens[0],
		(*C.float)(unsafe.Pointer(&flat[0]))); rc != C.EHX_OK {
		return t.fail()
	}
	return nil
}

// OnlineStoreTable.Get (online.go:52): returns exactly what was Set (vectorstore_test.go:107-113).
func (t *mi355xTable) Get(entity string) (interface{}, error) {
	defer pinned()()
	out := make([]float32, t.dims)
	ckey := C.CString(entity)
	defer C.free(unsafe.Pointer(ckey))
	switch rc := C.ehx_get(t.sp, ckey, C.size_t(len(entity)), (*C.float)(unsafe.Pointer(&out[0]))); rc {
	case C.EHX_OK:
		return out, nil
	case C.EHX_ENOTFOUND:
		return nil, fferr.NewEntityNotFoundError(t.feature, t.variant, entity, nil) // redis.go:445
	default:
		return nil, t.fail()
	}
}

// VectorStoreTable.Nearest (online.go:63): best first; returns <= k entities like Redis
// (redis.go:463-471), not Pinecone's ""-padded slice (pinecone.go:368-371).
func (t *mi355xTable) Nearest(feature, variant string, vector []float32, k int32) ([]string, error) {
	if len(vector) != t.dims || k <= 0 {
		return nil, fferr.NewInvalidArgumentError(fmt.Errorf("expected a %d-dim vector and k > 0", t.dims))
	}
	defer pinned()()
	ids := make([]C.uint64_t, k)
	dist := make([]C.float, k)
	off := make([]C.uint64_t, k+1)
	var count C.uint32_t
	arena := make([]byte, 64*int(k)+4096)
	for {
		rc := C.ehx_knn_keys(t.sp, 1, (*C.float)(unsafe.Pointer(&vector[0])), C.uint32_t(k), &ids[0], &dist[0], &count,
			(*C.char)(unsafe.Pointer(&arena[0])), C.size_t(len(arena)), &off[0])
		if rc == C.EHX_ERANGE {
			arena = make([]byte, 4*len(arena))
			continue
		}
		if rc != C.EHX_OK {
			return nil, t.fail() // redis.go:461
		}
		break
	}
	out := make([]string, int(count))
	for j := range out {
		out[j] = string(arena[off[j]:off[j+1]])
	}
	return out, nil
}
