// ffmodel.h -- the reference's FFModel operator API for the DLRM path, re-built as a thin C++
// host layer over the kernel C-ABI (include/ff_hip.h).  One process drives one GPU; there is no
// Legion, no mapper and no strategy search: placement is the fixed sharding of DESIGN.md
// (tables table-wise over ranks, everything else data-parallel).
//
// Mirrors, with the same names, argument meaning and error behaviour (print + abort):
//   FFConfig                      [ref: include/config.h:98-154, src/runtime/model.cc:2211-2403]
//   Tensor / Parameter            [ref: include/tensor.h:27-73, src/runtime/model.cu:337-467]
//   Initializer family            [ref: include/initializer.h:25-110]
//   Optimizer / SGDOptimizer      [ref: include/optimizer.h:30-60, src/runtime/optimizer.cc:43-189]
//   Op, Linear, Embedding, Concat, BatchMatmul [ref: include/model.h:205-271,968-1202,1739-1791]
//   FFModel                       [ref: include/model.h:283-588, src/runtime/model.cc:1410-1819]
//   PerfMetrics                   [ref: src/metrics_functions/metrics_functions.cc:20-80]
#pragma once

#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ff_hip.h"
#include "ffcomm.h"

// ---- enums: identical values to [ref: include/ffconst.h:4-57] -------------------------------
enum ActiMode { AC_MODE_NONE = 10, AC_MODE_RELU = 11, AC_MODE_SIGMOID = 12, AC_MODE_TANH = 13, AC_MODE_GELU = 14 };
enum AggrMode { AGGR_MODE_NONE = 20, AGGR_MODE_SUM = 21, AGGR_MODE_AVG = 22 };
enum DataType { DT_FLOAT = 40, DT_DOUBLE = 41, DT_INT32 = 42, DT_INT64 = 43, DT_BOOLEAN = 44, DT_NONE = 49 };
enum LossType {
  LOSS_CATEGORICAL_CROSSENTROPY = 50,
  LOSS_SPARSE_CATEGORICAL_CROSSENTROPY = 51,
  LOSS_MEAN_SQUARED_ERROR_AVG_REDUCE = 52,
  LOSS_MEAN_SQUARED_ERROR_SUM_REDUCE = 53,
};
enum CompMode { COMP_MODE_TRAINING = 70, COMP_MODE_INFERENCE = 71 };
enum ParameterSyncType { NONE = 80, PS = 81, NCCL = 82 };
enum MetricsType {
  METRICS_ACCURACY = 1001,
  METRICS_CATEGORICAL_CROSSENTROPY = 1002,
  METRICS_SPARSE_CATEGORICAL_CROSSENTROPY = 1004,
  METRICS_MEAN_SQUARED_ERROR = 1008,
  METRICS_ROOT_MEAN_SQUARED_ERROR = 1016,
  METRICS_MEAN_ABSOLUTE_ERROR = 1032,
};
enum OperatorType { OP_INPUT, OP_LINEAR, OP_EMBEDDING, OP_CONCAT, OP_BATCHMATMUL, OP_TRANSPOSE, OP_RESHAPE, OP_FLAT, OP_TRIL, OP_DOT_INTERACTION };

#define MAX_TENSOR_DIM 4
#define MAX_NUM_INPUTS 256
#define MAX_OPNAME 64

class FFModel;
class Op;
struct KernelApi;   // dlopen'ed table of the ffh_* entry points (backend.h)

// ---------------------------------------------------------------------------------------------
// ParallelConfig + strategy files [ref: include/config.h:47-73, src/runtime/strategy.cc:95-189].
// Text format: "<n_ops>" then per op "<name> <device_type> <nDims> <dim[0..nDims)> <n_ids> <ids...>", dims in
// Legion order (dim[nDims-1] = sample dim).  What this build can place: an Embedding on ONE device
// (all dims 1, device_ids[0] = owning rank -- the shape examples/cpp/DLRM/strategies/dlrm_strategy.cc:92-110 emits);
// every other op data-parallel over all ranks (only the sample dim split, ids 0..world_size-1).
struct ParallelConfig {
  enum DeviceType { GPU = 0, CPU = 1 };
  ParallelConfig() : device_type(GPU), nDims(0) { for (int& d : dim) d = 1; }
  int num_parts() const { int n = 1; for (int i = 0; i < nDims; i++) n *= dim[i]; return n; }
  bool is_data_parallel() const;      // only the sample dim is split
  bool operator==(const ParallelConfig& rhs) const;
  DeviceType device_type;
  int nDims, dim[MAX_TENSOR_DIM];
  std::vector<int> device_ids;
};
typedef size_t MappingTagID;
bool load_strategies_from_file(const std::string& filename, std::map<MappingTagID, ParallelConfig>& strategies);
bool save_strategies_to_file(const std::string& filename, const std::map<std::string, ParallelConfig>& strategies);

// ---------------------------------------------------------------------------------------------
class FFConfig {
 public:
  FFConfig();
  void parse_args(char** argv, int argc);
  // reference fields
  int epochs, batchSize, printFreq;
  int numNodes, cpusPerNode, workersPerNode;
  float learningRate, weightDecay;
  size_t workSpaceSize;
  bool syntheticInput, profiling, perform_fusion;
  CompMode computationMode;
  std::string dataset_path;
  std::string import_strategy_file, export_strategy_file;   // --import / --export [ref: src/runtime/model.cc:2327-2334]
  std::map<MappingTagID, ParallelConfig> strategies;        // keyed by get_hash_id(op name), as the reference
  static MappingTagID get_hash_id(const std::string& pcname);
  bool find_parallel_config(int ndims, const std::string& pcname, ParallelConfig& config) const;   // false: not in the file
  // this build
  std::string backend_lib;     // library exporting include/ff_hip.h; default: libffhip.so next to libffmodel.so
  int device;                  // HIP device ordinal of this process
  uint64_t seed;               // counter-based RNG seed (reference: unseeded std::rand)
  bool enable_graph;           // begin_trace/end_trace capture + replay as a hipGraph
  bool overlap_embedding;      // embedding gather (+exchange) on a side stream beside the bottom MLP
  bool dense_embedding_update; // reference's dense zero/scatter/sweep path instead of the fused sparse update
  int64_t column_shard_rows;   // tables with at least this many rows are sharded column-wise over the ranks (0: never)
  bool deterministic;          // --deterministic: weight / bias gradients without fp atomics (ffh_ctx_set_deterministic): bit-identical runs
  bool fp32_split_bf16x3;      // --fp32-split-bf16x3: wide Linear GEMMs fp32-accurate on the bf16 pipe (FFH_MATH_FP32_SPLIT_BF16X3)
  bool allow_tensor_op_math_conversion;   // --allow-tensor-op-math-conversion: bf16-operand MFMA GEMMs for the wide Linear layers (ffh_ctx_set_math_mode)
  bool bf16_convert_twins;               // tensor-op mode: a twin by conversion behind an fp32-kernel Linear that feeds a bf16-pipe one (--no-bf16-convert-twins)
  bool bf16_exact_small_backward;        // tensor-op mode: the backward of a small layer with a live activation derivative in exact mode (--no-bf16-exact-small-backward)
  bool bf16_twins, force_async_launch;   // --no-bf16-twins / --force-async-launch (A/B and test switches; they used to be environment variables)
  bool capture_exchange;             // --capture-exchange: world_size > 1 with collectives enqueued from C++ (RcclComm): the step is captured / replayed as a hipGraph
  bool pad_linear_k;                 // (A/B: --no-pad-linear-k) zero-pad the input / kernel of a wide Linear whose in_dim is not a multiple of 64
  bool sparse_embedding_optimizer;   // --sparse-embedding-optimizer: momentum / weight-decay SGD and Adam update the rows a batch touched, with per-row state, on the
                               // sorted segments of the fused update (ffh_sparse_opt: lazy semantics, a stated divergence) instead of the reference's dense sweep
  int early_sort;              // the index-only sort of the fused table update runs behind the gather (ffh_embedding_bwd_sort_multi), off the backward's critical path:
                               // 1 / 0 (--early-sort / --no-early-sort), -1 = by shape (FFModel::early_sort_possible)
  bool dx_colsum;              // a layer's bias gradient from the epilogue of the data-gradient kernel of the layer above (A/B: --no-dx-colsum)
  bool dx_scatter;             // exchange mode: the layer above the feature Concat writes its dX into the send buffer itself (A/B: --no-dx-scatter)
  bool fuse_pair;              // two narrow layers' backward as one launch + the lower dW GEMM (A/B: --no-fused-pair)
  int  bucket_allreduce;       // the MLP gradients' all-reduce in buckets issued from inside backward() on a communication stream, one per wide layer, beside the
                               // rest of the backward [ref: one ncclAllReduce per parameter from its own update task, src/runtime/optimizer.cc:93-189]:
                               // 1 / 0 (--bucket-allreduce / --no-bucket-allreduce), -1 = where the transport only enqueues (ffcomm.nonblocking)
  bool direct_allreduce;             // --direct-allreduce: a bucket's sum as all-to-all of 1 / N slices + local sum in rank order + all-gather (every xGMI link
                                     // carries 1 / N of it) instead of ncclAllReduce, whose ring one link bounds
  int64_t allreduce_bucket_floats;   // a bucket is closed once it holds this many gradients (--allreduce-bucket-floats N; default 1 Mi = 4 MB)
  int64_t big_dw_min_weights;  // ... only a layer with at least this many weights is cut (--big-dw-min-weights N; default 2 Mi)
  int  big_dw_chunks;          // with bucketed all-reduce: the biggest layer's weight-gradient GEMM as this many launches over row blocks of dW, a bucket behind each (its
                               // gradients are two thirds of the bytes and the last to be complete: --big-dw-chunks N; 0 / 1 = not cut, the default: see allocate step 5b)
  int  trace_mode;             // begin_trace / end_trace: 1 = every traced step is replayed from its hipGraph, as the reference traces every iteration
                               // [ref: examples/cpp/DLRM/dlrm.cc:174-181] (--always-replay); 0 = replayed only where that is not slower than launching it,
                               // measured on the trace's first calls, one GPU (--adaptive-replay): on this runtime the replay of a two-stream step costs a
                               // small model more than its launches (Kaggle shape: 210 vs 170 us).  -1 = not given: the FFModel API replays (1), the DLRM
                               // driver's timed loop adapts (0)
  bool mlp_chain;              // a run of narrow Linear layers (every width <= 512) as one launch forward, two backward (ffh_mlp_chain_fwd / _bwd; A/B: --no-mlp-chain)
  int64_t mlp_chain_max_batch; // ... for at most this many samples per GPU (--mlp-chain-max-batch N)
  int64_t mlp_chain_fwd_max_batch;   // ... and up to this many (--mlp-chain-fwd-max-batch N)
  int64_t mlp_chain_fwd_min_batch;   // the forward chain from this many samples per GPU up (below, the per-layer kernels win: --mlp-chain-fwd-min-batch N)
  int64_t mlp_chain_max_weights;     // chains of at most this many weights in all (every CU streams all of them from L2: --mlp-chain-max-weights N)
  bool attach_events;          // hang ev_grad_ready on the producing kernel's completion instead of a record packet (A/B: --no-attach-event)
  bool timing_events;          // A/B: stream-ordering events created with timestamps, as before
  bool fuse_loss;              // loss step + metrics inside the last layer's one-launch backward (A/B: --no-fused-loss)
  int64_t replicate_embedding_rows;   // world_size > 1: tables with at most this many rows are data-parallel (replicated) instead of owned by one rank (0: none)
  int64_t row_shard_rows;      // ... row-wise instead: partial bag sums + reduce-scatter (0: never; wins over column_shard_rows)
  bool async_launch;           // auxiliary streams are fed by their own host threads (HIP backend only)
  bool parallel_dw;            // weight-gradient GEMMs on their own stream beside the data-gradient chain
  bool force_exchange;         // run the all-to-all / all-reduce path even with one rank (tests the collectives on 1 GPU)
  ffcomm comm;                 // rank / world_size / collectives supplied by the launcher (ffcomm.h)
};

// ---------------------------------------------------------------------------------------------
struct TensorPiece {            // one column block of a tensor that is scattered over several buffers
  float*  ptr;
  float*  grad;
  int64_t ld;
  int     cols;
};

struct TensorImpl {
  void*   ptr = nullptr;        // device address of element (0, 0)
  int64_t ld = 0;               // elements between consecutive rows (rows = product of outer dims)
  float*  grad = nullptr;
  int64_t grad_ld = 0;
  bool    alias = false;        // lives inside another tensor's buffer (e.g. the concat output)
  bool    grad_alias = false;
  bool    is_input = false;
  size_t  bytes = 0;
  int64_t rows_local = 0;      // rows held by this rank (batch-sharded tensors: rows / world_size)
  int     guid = -1;
  // non-empty: the tensor has no single buffer; its columns are these pieces in order (the output of a
  // column-sharded embedding table inside the all-to-all receive buffer)
  std::vector<TensorPiece> pieces;
};

struct Tensor {
  Tensor(void);
  size_t get_volume() const;
  int64_t rows() const;         // product of all dims but the innermost
  int64_t cols() const { return adim[0]; }
  // host <-> device copies of the whole tensor, row-major with the batch outermost
  // [ref: src/runtime/model.cu:337-467 set_tensor/get_tensor]
  template <typename T> bool set_tensor(const FFModel* model, const std::vector<int>& dims, const T* data);
  template <typename T> bool get_tensor(const FFModel* model, T* data) const;
  template <typename T> bool get_grad(const FFModel* model, T* data) const;
  int numDim, adim[MAX_TENSOR_DIM];   // Legion order: adim[0] is the innermost dimension
  DataType data_type;
  ParameterSyncType sync_type;
  Op* owner_op;
  int owner_idx;
  TensorImpl* impl;                   // filled by FFModel::compile
};

struct Parameter : Tensor {
  template <typename T> bool set_weights(const FFModel* model, const std::vector<int>& dims, const T* data);
  template <typename T> bool get_weights(const FFModel* model, T* data) const;
};

// ---------------------------------------------------------------------------------------------
class Initializer {
 public:
  virtual ~Initializer() {}
  virtual void init(const FFModel* ff, const Parameter* p) = 0;
};
class ZeroInitializer : public Initializer {
 public:
  void init(const FFModel* ff, const Parameter* p) override;
};
class ConstantInitializer : public Initializer {
 public:
  explicit ConstantInitializer(float v) : value(v) {}
  void init(const FFModel* ff, const Parameter* p) override;
  float value;
};
class UniformInitializer : public Initializer {
 public:
  UniformInitializer(int _seed, float _min, float _max) : seed(_seed), min_val(_min), max_val(_max) {}
  void init(const FFModel* ff, const Parameter* p) override;   // device kernel, counter-based
  int seed;
  float min_val, max_val;
};
class NormInitializer : public Initializer {
 public:
  NormInitializer(int _seed, float _mean, float _stddev) : seed(_seed), mean(_mean), stddev(_stddev) {}
  void init(const FFModel* ff, const Parameter* p) override;   // host Box-Muller from the counter RNG, then upload
  int seed;
  float mean, stddev;
};
class GlorotUniform : public Initializer {
 public:
  explicit GlorotUniform(int _seed) : seed(_seed) {}
  void init(const FFModel* ff, const Parameter* p) override;
  int seed;
};

// ---------------------------------------------------------------------------------------------
class Optimizer {
 public:
  explicit Optimizer(const FFModel* _model) : model(_model) {}
  virtual ~Optimizer() {}
  virtual void init(void) = 0;
  virtual void next(void) = 0;
  virtual void update(const Parameter* p) = 0;
  const FFModel* model;
};
class SGDOptimizer : public Optimizer {
 public:
  SGDOptimizer(const FFModel* _model, double lr = 0.01f, double momentum = 0.0f, bool nesterov = false,
               double weight_decay = 0.0f);
  void init(void) override;
  void next(void) override;
  void update(const Parameter* p) override;
  void set_weight_decay(double wd) { weight_decay = wd; }
  double lr, momentum;
  bool nesterov;
  double weight_decay;
  std::map<const void*, float*> v_values;   // momentum buffers, keyed by weight pointer
};

// AdamOptimizer [ref: include/optimizer.h:62-85, src/runtime/optimizer.cc:190-330].  m and v live in two slabs laid
// out like the MLP parameter slab (one launch for every Linear parameter) plus one pair per dense embedding table.
class AdamOptimizer : public Optimizer {
 public:
  AdamOptimizer(const FFModel* _model, double _alpha = 0.001f, double _beta1 = 0.9f, double _beta2 = 0.999f,
                double _weight_decay = 0.0f, double _epsilon = 1e-8);
  void init(void) override;
  void next(void) override;
  void update(const Parameter* p) override;
  void set_weight_decay(double wd) { weight_decay = wd; }
  double alpha, beta1, beta2, weight_decay, epsilon;
  double alpha_t, beta1_t, beta2_t;
  float *mlp_m, *mlp_v;                                        // moments of the MLP slab
  std::map<const void*, std::pair<float*, float*>> mv_values;  // (m, v) of every other parameter, keyed by weight pointer
};

// ---------------------------------------------------------------------------------------------
struct PerfMetrics {
  PerfMetrics();
  void update(const PerfMetrics& one);
  void print(int flags) const;       // the reference's "[Metrics] ..." line on stderr
  int train_all, train_correct;
  float cce_loss, sparse_cce_loss, mse_loss, rmse_loss, mae_loss;
  double start_time;
};

// ---------------------------------------------------------------------------------------------
class Op {
 public:
  Op(FFModel& model, OperatorType type, const char* name, int num_inputs, const Tensor* inputs);
  virtual ~Op() {}
  virtual void init(const FFModel&) {}
  virtual void forward(const FFModel&) = 0;
  virtual void backward(const FFModel&) = 0;
  virtual void create_weights(FFModel&) {}
  virtual void create_output_and_partition(FFModel&) = 0;
  virtual void print_layer(const FFModel&) const;
  OperatorType op_type;
  char name[MAX_OPNAME];
  Tensor inputs[MAX_NUM_INPUTS];
  Tensor outputs[1];
  Parameter weights[2];
  int numInputs, numWeights, numOutputs;
  int layer_index;              // position in FFModel::layers
  bool profiling;
};

class Linear : public Op {
 public:
  Linear(FFModel& model, const Tensor& input, int out_dim, ActiMode activation, bool use_bias, const Op* shared_op,
         Initializer* kernel_initializer, Initializer* bias_initializer, const char* name);
  void create_weights(FFModel& model) override;
  void create_output_and_partition(FFModel& model) override;
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  void backward_part(const FFModel&, int part);   // 0: all; 1: data gradient only
  void backward_dw_rows(const FFModel&, int row0, int nrows);   // the weight (+ bias) gradient of output rows [row0, row0 + nrows) on the weight-gradient stream (dy final: premasked / no activation)
  int in_channels, out_channels;
  int in_padded;                // what the kernel library is told: in_channels, or that rounded up to 64 when the input tensor and the kernel were
                                // given zero pad columns (FFModel::allocate step 4a: reduction depths the persistent GEMMs cannot take)
  ActiMode activation;
  bool use_bias;
  bool discard_input_grad;      // first layer on a model input: dX is never consumed
  bool dx_overwrite;            // input has no other consumer: dX may be stored instead of accumulated
  ffh_col_dest* dx_map;         // device array [in_channels]: where each column of dX goes when the Concat below is folded in (exchange mode)
  class Concat* dx_map_concat;  // ... and the Concat whose backward that replaces
  Linear* pair_upper;           // the narrow layer above, when its forward rides in this layer's launch (ffh_linear_pair_fwd)
  mutable bool fwd_done_by_pair;   // set by the layer below for this forward()
  Linear* pair_lower;           // the layer below, when its data gradient rides in this layer's backward launch (ffh_linear_pair_bwd)
  int backward_pair(const FFModel&);   // FFH_OK: this layer's backward and the lower layer's whole backward are enqueued
  bool dx_mask_by_x, dy_premasked;   // relu' of the layer below applied by this layer's dX epilogue / already applied by the layer above
  Linear* colsum_lower;         // the Linear below whose FINAL dy is the dX this layer stores: its bias gradient can come out of this layer's
                                // data-gradient kernel (ffh_linear_bwd_set_dx_colsum, ABI 10) instead of riding on its own weight-gradient GEMM
  bool db_from_upper;           // set by the layer above for this backward(): the bias gradient is done, the call passes db = NULL
  std::vector<Linear*> chain_fwd;  // non-empty on the LOWEST layer of a chain of narrow layers: its members bottom -> top; forward() of that layer
                                   // launches all of them (ffh_mlp_chain_fwd, ABI 12)
  mutable bool fwd_done_by_chain;  // set by the chain's lowest layer for this forward()
  void* out_twin = nullptr;        // tensor-op mode: where forward() leaves the bf16 rounding of its output (allocate() step 7), or null
  bool bwd_exact = false;          // tensor-op mode: backward() runs this layer's two GEMMs in exact mode (allocate() step 7: a small layer whose dy arrives with a live activation derivative)
  void* dx_twin = nullptr;         // ... and then refreshes the bf16 twin of the data gradient it stored, where that buffer has one
  bool dx_twin_registered = false;
  bool dx_image = false;           // split mode: backward() leaves the three-plane image of the data gradient it stores (allocate() step 7)
  bool out_twin_x3 = false;        // ... split mode: out_twin is the output's three-plane image (ffh_convert_f32_to_bf16x3 finds it by the registration)
  std::vector<Linear*> chain_bwd;  // non-empty on the TOP layer of the chain FFModel::backward runs as one call (ffh_mlp_chain_bwd): members bottom -> top
  Initializer *kernel_initializer, *bias_initializer;
};

class Embedding : public Op {
 public:
  Embedding(FFModel& model, const Tensor& input, int num_entries, int outDim, AggrMode aggr, const Op* shared_op,
            Initializer* kernel_initializer, const char* name);
  void create_weights(FFModel& model) override;
  void create_output_and_partition(FFModel& model) override;
  void forward(const FFModel&) override;     // first table of a group launches the whole group
  void backward(const FFModel&) override;
  int num_entries, out_channels;
  AggrMode aggr;
  Initializer* kernel_initializer;
  int table_index;              // position among the model's embedding ops
  int owner_rank;               // table-wise sharding: table_index % world_size
  bool column_sharded;          // every rank holds out_channels / world_size columns of all rows
  int local_cols;               // columns of the table held by this rank
  // row-wise sharding (--row-shard-rows): this rank holds rows [row_begin, row_begin + rows_local) plus one all-zero row
  // that stands for every row held elsewhere; forward = gather of partial bag sums for the GLOBAL batch + reduce-scatter,
  // backward = all-gather of the output gradients + fused update of the local rows
  bool row_sharded;
  int64_t row_begin, rows_local;
  int64_t* local_idx;           // [batch][bag] ids relative to row_begin (rows held elsewhere -> rows_local)
  float *partial, *gfull;       // [batch][out_channels]: partial sums (reduce-scatter input), gathered gradients
  void set_row_sharding(const FFModel& model, bool on);
  // data-parallel table (the reference's DEFAULT placement: an op without a strategy entry is split on the sample dim and its
  // weights replicated [ref: src/runtime/model.cc:500-510], gradient synchronised like any parameter's [ref: ncclAllReduce,
  // src/runtime/optimizer_kernel.cu:170-171]): every rank holds the whole table, gathers its own samples, scatter-adds a dense
  // gradient [ref: embed_backward, src/ops/embedding.cu:192-217]; table and gradient live in the dense parameter slab, so the
  // MLP all-reduce bucket and the slab SGD / Adam launch cover them.  --replicate-embedding-rows N or a strategy file.
  bool replicated;
  void set_replicated(const FFModel& model, bool on);
  float* opt_state[2] = {nullptr, nullptr};   // --sparse-embedding-optimizer: per-row state of this rank's slice (SGD momentum: V; Adam: M, V)
  bool held_here(int rank) const { return owner_rank == rank || column_sharded || row_sharded || replicated; }
};

class Concat : public Op {
 public:
  Concat(FFModel& model, int n, const Tensor* inputs, int axis, const char* name);
  void create_output_and_partition(FFModel& model) override;
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  mutable bool bwd_done;        // the consumer stored its data gradient straight into this Concat's inputs (ffh_linear_bwd_set_dx_scatter)
  int axis;                     // Legion axis (user axis flipped, [ref: src/ops/concat.cu:29-49,109-112])
  bool bwd_overwrite;           // every input has this Concat as its only consumer: slices are stored, not accumulated
  std::vector<int> image_inputs;   // split mode: inputs written in place by layers that keep no image (fp32-kernel Linears): forward() converts their slices
};

class BatchMatmul : public Op {
 public:
  BatchMatmul(FFModel& model, const Tensor& A, const Tensor& B, int a_seq_length_dim, int b_seq_length_dim);
  void create_output_and_partition(FFModel& model) override;
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  int a_seq_length_dim, b_seq_length_dim;
};

// The three shape ops the reference composes the dot interaction from
// [ref: src/ops/transpose.cu, src/ops/reshape.cu:203-210, src/ops/flat.cu:117-124]
class Transpose : public Op {
 public:
  Transpose(FFModel& model, const Tensor& input, const std::vector<int>& perm, const char* name);
  void create_output_and_partition(FFModel&) override {}
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  int perm[MAX_TENSOR_DIM];     // natural order: output dim i = input dim perm[i]
};
class Reshape : public Op {     // also serves Flat: a copy forward, an accumulate backward
 public:
  Reshape(FFModel& model, OperatorType type, const Tensor& input, const std::vector<int>& shape, const char* name);
  void create_output_and_partition(FFModel&) override {}
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  bool is_view;                 // the output IS the input's buffer (contiguous input read by nothing else): no copy either way
};
// Strict lower triangle of [batch][n][n] -> [batch][n (n - 1) / 2]: the extraction MLPerf-DLRM's dot interaction applies to
// Z Z^T.  No reference operator (its dot interaction is a TODO, examples/cpp/DLRM/dlrm.cc:53-54); parity is against torch.
class Tril : public Op {
 public:
  Tril(FFModel& model, const Tensor& input, const char* name);
  void create_output_and_partition(FFModel&) override {}
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  int n;
};
// The pairwise-dot interaction in one launch each way (csrc/interaction.hip): input [batch][c * d] (the Concat of the
// bottom-MLP output and the c - 1 embedding outputs), output [batch][d + c (c - 1) / 2] = [row 0 | <z_i, z_j>, i > j] --
// what concat -> reshape -> transpose -> batch_matmul -> tril -> concat(x, .) computes with 13 passes over the block.
class DotInteraction : public Op {
 public:
  DotInteraction(FFModel& model, const Tensor& input, int d, const char* name);
  void create_output_and_partition(FFModel&) override {}
  void forward(const FFModel&) override;
  void backward(const FFModel&) override;
  int c, d;
  bool bwd_overwrite;           // the input has no other consumer: its gradient is stored, not accumulated
};

// ---------------------------------------------------------------------------------------------
// A host thread that issues the launches of one auxiliary HIP stream.  The training step is bound by
// the host's launch rate (~7 us per launch, ~35 launches per Kaggle step); the weight-gradient GEMMs
// and the embedding side stream are therefore enqueued by their own threads while the main thread
// walks the critical path.  Ordering between streams is by HIP events; the only host-side rule is
// that a thread may wait on an event only after the recording thread has issued the record, which
// `drain()` provides at the few join points.  (What Legion's utility processors do for the reference.)
class LaunchWorker {
 public:
  LaunchWorker(const KernelApi* api, int device);
  ~LaunchWorker();
  void post(std::function<void(ffh_ctx*)> fn);
  void drain();
  ffh_ctx* ctx() const { return wctx; }
 private:
  void run();
  const KernelApi* api;
  int device;
  ffh_ctx* wctx;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_idle;
  std::deque<std::function<void(ffh_ctx*)>> q;
  bool stop, busy;
};

// ---------------------------------------------------------------------------------------------
class FFModel {
 public:
  explicit FFModel(FFConfig& config);
  ~FFModel();

  Tensor embedding(const Tensor& input, int num_entires, int outDim, AggrMode aggr, const Op* shared_op = NULL,
                   Initializer* kernel_initializer = NULL, const char* name = NULL);
  Tensor batch_matmul(const Tensor& A, const Tensor& B, int a_seq_length_dim = -1, int b_seq_length_dim = -1);
  Tensor dense(const Tensor& input, int outDim, ActiMode activation = AC_MODE_NONE, bool use_bias = true,
               const Op* shared_op = NULL, Initializer* kernel_initializer = NULL,
               Initializer* bias_initializer = NULL, const char* name = NULL);
  Tensor concat(int n, const Tensor* tensors, int axis, const char* name = NULL);
  Tensor flat(const Tensor& input, const char* name = NULL);
  Tensor tril(const Tensor& input, const char* name = NULL);
  Tensor dot_interaction(const Tensor& input, int d, const char* name = NULL);   // fused pairwise-dot interaction (this build's addition)   // strict lower triangle of [batch][n][n] (this build's addition)
  Tensor transpose(const Tensor& input, const std::vector<int>& perm, const char* name = NULL);
  Tensor reshape(const Tensor& input, const std::vector<int>& shape, const char* name = NULL);
  template <int NDIM>
  Tensor create_tensor(const int dims[], DataType data_type, const Op* owner_op = NULL, bool create_grad = true);
  template <int NDIM>
  Parameter create_weight(const int dims[], const Op* op, DataType data_type, Initializer* initializer,
                          bool create_grad = true);

  void compile(LossType loss_type, const std::vector<MetricsType>& metrics, CompMode comp_mode = COMP_MODE_TRAINING);
  void compile(Optimizer* optimizer, LossType loss_type, const std::vector<MetricsType>& metrics,
               CompMode comp_mode = COMP_MODE_TRAINING);
  void init_layers();
  void reset_metrics();
  void forward(int seq_length = -1);
  void zero_gradients();
  void compute_metrics();
  void backward(int seq_length = -1);
  void update();
  // Legion trace memoisation [ref: examples/cpp/DLRM/dlrm.cc:174-181] == hipGraph capture + replay
  void begin_trace(int trace_id);
  void end_trace(int trace_id);
  void sync();                                   // issue_execution_fence + wait
  PerfMetrics get_perf_metrics();
  // seeds of the initializers: a private counter-hash sequence from config.seed (the reference draws them from the
  // unseeded global std::rand(), [ref: examples/cpp/DLRM/dlrm.cc:32,34,45] -- any library calling rand() would shift it)
  Initializer* own(Initializer* init) { owned_initializers.push_back(init); return init; }   // deleted with the model (the reference leaks these)
  int next_seed();
  uint64_t seed_counter;                // device -> host (synchronises)
  void print_layers(int id);
  std::string get_operator_type_name(OperatorType type) const;

  // ---- state ---------------------------------------------------------------------------------
  int op_global_guid;
  FFConfig config;
  Optimizer* optimizer;
  LossType loss_type;
  int metrics_flags;
  Tensor label_tensor;
  std::vector<Op*> layers;
  std::vector<Parameter> parameters;
  int seq_length;

  // ---- runtime (what FFHandler + Legion regions are in the reference) ------------------------
  const KernelApi* api;
  ffh_ctx* ctx;
  ffh_stream stream;           // main compute stream
  ffh_stream side_stream;      // embedding gather / exchange / sparse update
  ffh_stream dw_stream;        // weight-gradient GEMMs (parallel_dw): the biggest layer's ...
  ffh_event ev_dw_done;
  int big_dw_layer;            // the Linear with the most multiply-adds
  bool need_zero_gsend;        // some gradient in the exchange send buffer is accumulated rather than stored
  bool need_zero_act_grads;    // some activation gradient is accumulated by more than one producer
  mutable bool dw_forked;
  mutable bool dw1_used = false;   // this step's forks were offered the weight-gradient stream (joined in update())
  mutable bool dw_stream_used_directly = false;   // a weight gradient was enqueued on dw_stream by this layer itself (deferred dW), not by the library's fork
  struct GradBucket {
    size_t off, count;          // floats in the dense gradient slab
    int lowest_layer;           // complete once every layer >= this one has issued its backward
    bool issued;
    ffh_event ready, ready_dw, done;
    int chunk_layer, chunk_index;   // >= 0: the bucket of one row block of that layer's weight gradient (FFConfig::big_dw_chunks), issued by the layer's own backward
    bool inline_issued;             // ran on the compute stream itself (capture): nothing to join
  };
  std::vector<GradBucket> grad_buckets;                 // in the order the backward completes them (top layers first)
  std::vector<std::pair<size_t, size_t>> grad_rest;     // (offset, count) of what no bucket covers (data-parallel tables): reduced in update()
  ffh_stream ar_stream = nullptr;                       // the buckets' stream
  bool bucketed_now() const;                            // buckets are issued from backward() in this step
  int  allreduce_grads(float* buf, int64_t count, ffh_stream s, bool bucket) const;   // ring (the transport's all-reduce) or direct; 0 = ok
  float* ar_scratch = nullptr;          // direct all-reduce: the received slices + the gathered sums (2 x the largest bucket, padded)
  size_t ar_scratch_floats = 0;
  mutable int64_t n_direct_allreduces = 0;
  mutable std::map<int64_t, std::vector<int64_t>> direct_plans;   // count -> [send counts | receive counts] of its all-to-all (stable addresses)
  bool buckets_held() const;          // a shared channel and this step's backward all-to-all not enqueued yet
  void issue_grad_buckets(int next_layer);              // every complete, not yet issued bucket (layers > next_layer have issued their backward)
  void issue_one_bucket(size_t k, bool wait_main);      // wait_main: also behind what the compute stream holds now
  int  big_dw_chunks_now() const;                       // row blocks the biggest layer's weight gradient is cut into this step (1: not cut)
  mutable int64_t n_bucket_allreduces = 0;
  mutable int64_t n_chain_fwd_calls = 0, n_chain_bwd_calls = 0;   // successful ffh_mlp_chain_fwd / _bwd calls (tests: flexflow_model_get_counter)
  bool mlp_chain_usable(int64_t rows, bool fwd) const;      // the chain launches are allowed in this mode / at this batch
  int run_chain_fwd(const Linear* lowest) const;  // FFH_OK, or FFH_ERR_UNSUPPORTED with nothing launched
  int run_chain_bwd(Linear* top);
  mutable bool mlp_grads_clean;   // the optimizer kernel cleared the MLP gradient slab (FFH_OPT_ZERO_GRAD): zero_gradients() skips it
  LaunchWorker *dw_worker, *side_worker;   // NULL: launches are issued inline by the calling thread
  std::vector<ffh_event> layer_events;     // one per layer: "dY of this layer is ready"
  bool use_workers() const { return dw_worker != nullptr && capturing_trace < 0; }
  ffh_event ev_fork, ev_join, ev_grad_ready, ev_update_done;
  int rank, world_size;
  bool exchange;               // table-wise exchange + gradient all-reduce active (world_size > 1)
  int64_t local_batch;         // config.batchSize / world_size
  bool compiled;

  void* dmalloc(size_t bytes) const;
  void check(int rc, const char* what) const;   // non-zero rc -> print ffh_last_error_string + abort (reference behaviour)

  // embedding group (all Embedding ops share L, D, aggr in DLRM): batched launches + exchange
  std::vector<Embedding*> embeddings;
  void embedding_group_forward(ffh_stream s, ffh_ctx* on_ctx = nullptr) const;   // on_ctx: the issuing thread's ctx
  void replicated_embedding_grads() const;                     // dense gradient of the data-parallel tables into the slab (compute stream)
  void embedding_group_update(ffh_stream s, ffh_ctx* on_ctx = nullptr) const;
  // this rank's gather / fused update kernels, no exchange.  idx_override (bench probes): one id buffer per owned shard, in shard
  // order, used instead of the model's own -- back-to-back probe launches rotate over several id sets so that no launch finds the
  // rows of the launch before in the 256 MiB Infinity Cache
  void embedding_kernels_only(bool fwd, ffh_stream s, const std::vector<const int64_t*>* idx_override = nullptr) const;
  // bench probes: events around the side-stream gather / update of a REAL step (what the kernels take while they share the chip)
  mutable bool probe_events_on = false;
  // pairs: 0/1 gather (+ forward exchange), 2/3 table update (+ backward exchange), 4/5 forward all-to-all, 6/7 backward all-to-all,
  // 8/9 gradient all-reduce (compute stream), 10/11 the compute stream's wait for the gather / exchange branch (its exposed part)
  static constexpr int kProbeEvents = 30;     // pairs: gather, update, a2a fwd, a2a bwd, all-reduce (rest / single bucket), join wait, the compute stream's wait for the buckets, buckets 0..7
  mutable ffh_event probe_ev[kProbeEvents] = {};
  void probe_record(int which, ffh_stream s, ffh_ctx* cx) const;
  bool fused_embedding_update() const;        // the tables are updated on the sorted segments (plain SGD, or any optimizer with --sparse-embedding-optimizer)
  bool sparse_rule(ffh_sparse_opt& rule) const;   // the row rule in force; false: plain SGD
  mutable bool opt_next_done = false;          // Optimizer::next() has run for the step in flight (update() does it; backward() clears it)
  void embedding_dense_update() const;         // the reference's dense path on the holders of each table (any optimizer, any placement)
  void issue_embedding_forward_on_side_stream() const;
  void join_embedding_forward() const;
  void issue_embedding_update_on_side_stream() const;
  void order_input_writes_behind_update() const;
  void profiled(const Op* op, bool fwd, const std::function<void()>& fn) const;   // --profiling: one op between two events
  mutable bool emb_forward_issued, emb_forward_joined, emb_update_pending;
  static constexpr int early_sort_big_batch_mode = 0;   // one GPU, >= 8192 samples: 0 = the sort stays in front of the apply phase (see early_sort_possible)
  bool early_sort_possible(int where) const;
  mutable bool emb_sorted_early;         // this step's sort was issued behind the gather: the update is the apply phase only
  int scatter_attach_layer;     // exchange mode: the Linear whose scattered dX completes the embedding output gradients (-1: none)
  std::vector<Initializer*> owned_initializers;
  int grad_attach_layer;        // the Linear whose backward completes the embedding output gradients (-1: none / not attachable)
  // tensor-op math mode: bfloat16 twins of the weight slab, of activations and of activation gradients whose every writer
  // keeps a twin current (ffh_ctx_bf16_mirror_set): the bf16-pipe GEMMs then read 2-byte operands instead of rounding 4-byte ones
  void* w_twin = nullptr; void* act_twin = nullptr; void* grad_twin = nullptr;
  mutable bool w_twin_dirty = false;      // a host write / initializer touched the weights: reconvert before the next step
  void refresh_weight_twin() const;
  void note_weight_write(const void* p) const;
  mutable bool bwd_alltoall_issued = false;   // this step's backward all-to-all has been enqueued (FFModel::issue_grad_buckets: a shared channel holds the buckets until then)
  int n_twin_regions = 0;
  int z_reader_layer;           // the lowest-index Linear that reads a Concat output the tables are gathered into (-1: unknown): behind ITS
                                // backward no forked weight-gradient GEMM reads that buffer any more, so the next gather may overwrite it
  ffh_event ev_z_free;          // recorded on dw_stream behind that layer's backward
  mutable bool z_free_recorded;
  mutable bool grad_ready_attached;

  // slabs
  float *mlp_weights, *mlp_grads;  size_t mlp_count;          // all Linear params, contiguous (one all-reduce, one SGD launch)
  char* act_slab;                                             // every activation that needs its own storage
  char* act_grad_slab;  size_t act_grad_bytes;                // every activation gradient (one memset per step)
  void* workspace;  size_t workspace_bytes;
  void* repl_workspace;  size_t repl_workspace_bytes;          // scratch of the data-parallel tables' gradient (own buffer: it runs beside the side-stream update)
  ffh_perf_metrics* d_perf;
  // one unit of the exchange: a whole table (table-wise) or a column block of a giant table (column-wise)
  struct EmbShard { Embedding* e; int owner; int col0; int cols; int64_t off; };
  std::vector<EmbShard> shards;
  std::vector<int64_t> rank_width;                            // columns each rank contributes to one sample's row
  // table-wise exchange buffers (world_size > 1)
  float *xsend, *xrecv, *gsend, *grecv;
  std::vector<int64_t> fwd_send_counts, fwd_recv_counts;      // floats per peer
  std::vector<int> owned_tables;                              // table indices this rank owns
  int tables_of_rank(int r) const;

  // trace / graph
  std::map<int, ffh_graph> graphs;
  int capturing_trace, replaying_trace;
  struct TraceTune { int calls = 0; int decided = 0; ffh_event ev[4] = {nullptr, nullptr, nullptr, nullptr}; float eager_ms = 0.f, graph_ms = 0.f; };   // decided: 0 not yet, 1 replay, 2 eager
  std::map<int, TraceTune> trace_tune;
  bool trace_adaptive() const { return config.trace_mode == 0 && !exchange; }      // (-1 counts as 1 here; DLRMApp::run_epochs turns -1 into 0)
  bool trace_replays(int trace_id) const;      // this trace's steps are (or will be) replayed
  mutable bool inputs_dirty;    // an input tensor was written on `stream` since the last forward(): the side-stream gather must be ordered behind it
  mutable bool fork_recorded;   // this forward() recorded ev_fork

  std::vector<TensorImpl*> tensor_impls;
  std::vector<Tensor*> input_tensors;
 private:
  void allocate();
  void apply_strategies();      // --import / --export [ref: src/runtime/model.cc:1575-1577, src/runtime/simulator.cu:131-143]
};
