"""INTEGRATION.md section 2 is code a maintainer of the reference would paste: this test keeps it honest.

The stub block (between the `stubs:begin` / `stubs:end` markers of INTEGRATION.md) is compiled with `g++ -fsyntax-only`
against include/ff_hip.h behind a prelude that DECLARES the reference's types the bodies touch, with the static member
signatures exactly as the reference declares them (signatures only -- category (b) in the copy check):
  OpMeta / FFHandler           [ref: include/model.h:197-203, include/config.h:75-84]
  LinearMeta, Linear           [ref: include/model.h:968-977, 1011-1027]
  BatchMatmulMeta, BatchMatmul [ref: include/model.h:1070-1074, 1098-1118]
  Embedding, EmbeddingMeta     [ref: include/model.h:1169-1186, 1198-1202]
  Concat                       [ref: include/model.h:1771-1784]
  SGDOptimizer / AdamOptimizer fields [ref: include/optimizer.h:57-60, 85-86]
An `int` vs `int64_t`, a `const` or an argument-count mismatch between a stub and either side fails the compile.
"""
import os
import re
import subprocess

from conftest import ROOT

PRELUDE = r"""
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "ff_hip.h"

// ---- what the reference's headers provide (declarations only) ----
typedef struct ihipStream_t* cudaStream_t;           // the maintainer's build maps cudaStream_t to hipStream_t
typedef long long coord_t;                           // Legion's coord_t
#define MAX_NUM_INPUTS 256                           // include/config.h:30-37
struct Domain { int get_dim() const; };              // Legion::Domain (opaque here)
enum ActiMode { AC_MODE_NONE = 10, AC_MODE_RELU = 11, AC_MODE_SIGMOID = 12, AC_MODE_TANH = 13, AC_MODE_GELU = 14 };
enum AggrMode { AGGR_MODE_NONE = 20, AGGR_MODE_SUM = 21, AGGR_MODE_AVG = 22 };
struct FFHandler { void* workSpace; size_t workSpaceSize; bool allowTensorOpMathConversion; ffh_ctx* ffh; /* <- the one added field (INTEGRATION.md section 1) */ };
class OpMeta { public: OpMeta(FFHandler _handle); FFHandler handle; bool profiling; };
class LinearMeta : public OpMeta { public: LinearMeta(FFHandler handle, int batch_size); const float* one_ptr; ActiMode activation; bool use_bias; };
class BatchMatmulMeta : public OpMeta { public: BatchMatmulMeta(FFHandler handler); int a_seq_length_dim, b_seq_length_dim; };
class EmbeddingMeta : public OpMeta { public: EmbeddingMeta(FFHandler handle); AggrMode aggr; };
struct SGDOptimizer { double lr, momentum; bool nesterov; double weight_decay; };
struct AdamOptimizer { double alpha, beta1, beta2, weight_decay, epsilon; double alpha_t, beta1_t, beta2_t; };
// the body of Concat::forward_kernel up to its launches (calc_blk_size over the Legion rects, src/ops/concat.cu:194-240), kept as it is
void concat_blk_sizes(coord_t& num_blocks, coord_t& output_blk_size, coord_t* input_blk_sizes, int num_inputs, int axis,
                      const Domain& out_domain, const Domain* in_domain);

class Embedding {
public:
  static void forward_kernel(int64_t const *input_ptr,
                             float *output_ptr,
                             float const *weight_ptr,
                             int in_dim,
                             int out_dim,
                             int batch_size,
                             AggrMode aggr,
                             int outputSize,
                             cudaStream_t stream);
  static void backward_kernel(int64_t const *input_ptr,
                              float const *output_ptr,
                              float *weight_grad_ptr,
                              int in_dim,
                              int out_dim,
                              int batch_size,
                              AggrMode aggr,
                              int outputSize,
                              cudaStream_t stream);
};
class Linear {
public:
  static void forward_kernel(const LinearMeta* m,
                      const float* input_ptr,
                      float* output_ptr,
                      const float* filter_ptr,
                      const float* bias_ptr,
                      int in_dim, int out_dim, int batch_size,
                      cudaStream_t stream);
  static void backward_kernel(const LinearMeta* m,
                       const float* input_ptr,
                       float* input_grad_ptr,
                       const float* output_ptr,
                       float* output_grad_ptr,
                       const float* kernel_ptr,
                       float* kernel_grad_ptr,
                       float* bias_ptr,
                       int in_dim, int out_dim, int batch_size,
                       cudaStream_t stream);
};
class BatchMatmul {
public:
  static void forward_kernel(const BatchMatmulMeta* meta,
                      float* o_ptr,
                      const float* a_ptr,
                      const float* b_ptr,
                      const float* c_ptr,
                      int m, int n, int k,
                      int batch,
                      cudaStream_t stream,
                      int a_seq_length_dim = -1,
                      int b_seq_length_dim = -1,
                      int seq_length = -1);
  static void backward_kernel(const BatchMatmulMeta* meta,
                       const float* o_ptr,
                       const float* o_grad_ptr,
                       const float* a_ptr,
                       float* a_grad_ptr,
                       const float* b_ptr,
                       float* b_grad_ptr,
                       float* c_grad_ptr,
                       int m, int n, int k, int batch,
                       cudaStream_t stream);
};
class Concat {
public:
  static void forward_kernel(float* output,
                             float const * const *inputs,
                             int num_inputs,
                             int axis,
                             const Domain& out_domain,
                             const Domain* in_domain,
                             cudaStream_t stream);
  static void backward_kernel(const float* output_grad,
                              float** input_grads,
                              int num_inputs,
                              int axis,
                              const Domain& out_grad_domain,
                              const Domain* in_grad_domain,
                              cudaStream_t stream);
};
// ---- end of prelude: INTEGRATION.md's bodies follow ----
"""


def _stub_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- stubs:begin -->\s*```cpp\n(.*?)```\s*<!-- stubs:end -->", text, re.S)
    assert m, "INTEGRATION.md lost its stubs:begin / stubs:end block"
    return m.group(1)


def test_integration_stubs_compile_against_the_reference_signatures(tmp_path):
    src = tmp_path / "stubs.cc"
    src.write_text(PRELUDE + _stub_block())
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-Wno-unused-function",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_integration_stubs_cover_the_four_operator_statics():
    body = _stub_block()
    for name in ("Embedding::forward_kernel", "Embedding::backward_kernel", "Linear::forward_kernel", "Linear::backward_kernel",
                 "Concat::forward_kernel", "Concat::backward_kernel", "BatchMatmul::forward_kernel", "BatchMatmul::backward_kernel"):
        assert re.search(r"void\s+" + re.escape(name) + r"\s*\(", body), name
    # the two handle-less statics must not reach for a handle they cannot see
    for cls in ("Embedding", "Concat"):
        for fn in re.findall(r"void\s+" + cls + r"::\w+\s*\([^{]*\{(.*?)\n\}", body, re.S):
            assert "handle." not in fn and "m->" not in fn, f"{cls} static uses a handle it does not receive"


def test_a_signature_mismatch_is_caught(tmp_path):
    """The check has teeth: the same block with one argument narrowed must fail to compile."""
    bad = _stub_block().replace("ffh_linear_fwd(m->handle.ffh, input_ptr, in_dim, output_ptr", "ffh_linear_fwd(m->handle.ffh, output_ptr, in_dim, input_ptr", 1)
    assert bad != _stub_block()
    src = tmp_path / "bad.cc"
    src.write_text(PRELUDE + bad)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode != 0
