/* ffmodel_c.h -- extern "C" face of the FFModel shim, in the style of the reference's
 * python/flexflow_c.h (opaque {void* impl} handles, [ref: python/flexflow_c.h:24-42,108-138,
 * 196-248,498-546]).  Python (bench.py, run_dlrm.py, tests) binds it with ctypes.
 * Errors abort the process with a message, as the reference's asserts do. */
#ifndef FFMODEL_C_H_
#define FFMODEL_C_H_
#include <stdbool.h>
#include <stdint.h>
#include "ffcomm.h"
#ifdef __cplusplus
extern "C" {
#endif

#define FF_NEW_OPAQUE_TYPE(T) typedef struct T { void* impl; } T
FF_NEW_OPAQUE_TYPE(flexflow_config_t);
FF_NEW_OPAQUE_TYPE(flexflow_model_t);
FF_NEW_OPAQUE_TYPE(flexflow_tensor_t);
FF_NEW_OPAQUE_TYPE(flexflow_initializer_t);
FF_NEW_OPAQUE_TYPE(flexflow_sgd_optimizer_t);
FF_NEW_OPAQUE_TYPE(flexflow_adam_optimizer_t);
FF_NEW_OPAQUE_TYPE(flexflow_dlrm_t);
#undef FF_NEW_OPAQUE_TYPE

typedef struct flexflow_perf_metrics_t {
  int train_all, train_correct;
  float cce_loss, sparse_cce_loss, mse_loss, rmse_loss, mae_loss;
} flexflow_perf_metrics_t;

/* FFConfig */
flexflow_config_t flexflow_config_create(void);
void flexflow_config_destroy(flexflow_config_t);
void flexflow_config_parse_args(flexflow_config_t, char** argv, int argc);
void flexflow_config_set_comm(flexflow_config_t, const ffcomm* comm);
void flexflow_config_set_batch_size(flexflow_config_t, int);
int  flexflow_config_get_batch_size(flexflow_config_t);
void flexflow_config_set_backend(flexflow_config_t, const char* lib_path);
void flexflow_config_set_seed(flexflow_config_t, uint64_t);
void flexflow_config_set_device(flexflow_config_t, int);
void flexflow_config_set_enable_graph(flexflow_config_t, bool);
void flexflow_config_set_overlap_embedding(flexflow_config_t, bool);
void flexflow_config_set_dense_embedding_update(flexflow_config_t, bool);

/* FFModel */
flexflow_model_t flexflow_model_create(flexflow_config_t);
void flexflow_model_destroy(flexflow_model_t);
flexflow_tensor_t flexflow_tensor_create(flexflow_model_t, int num_dims, const int* dims, int data_type, bool create_grad);
flexflow_tensor_t flexflow_model_add_dense(flexflow_model_t, flexflow_tensor_t input, int out_dim, int activation, bool use_bias,
                                           flexflow_initializer_t kernel_init, flexflow_initializer_t bias_init, const char* name);
flexflow_tensor_t flexflow_model_add_embedding(flexflow_model_t, flexflow_tensor_t input, int num_entries, int out_dim, int aggr,
                                               flexflow_initializer_t kernel_init, const char* name);
flexflow_tensor_t flexflow_model_add_concat(flexflow_model_t, int n, const flexflow_tensor_t* inputs, int axis, const char* name);
flexflow_tensor_t flexflow_model_add_flat(flexflow_model_t, flexflow_tensor_t input, const char* name);
flexflow_tensor_t flexflow_model_add_dot_interaction(flexflow_model_t, flexflow_tensor_t input, int d, const char* name);   /* [batch][c*d] -> [batch][d + c(c-1)/2] */
flexflow_tensor_t flexflow_model_add_tril(flexflow_model_t, flexflow_tensor_t input, const char* name);   /* strict lower triangle of [batch][n][n] */
flexflow_tensor_t flexflow_model_add_transpose(flexflow_model_t, flexflow_tensor_t input, int n, const int* perm, const char* name);
flexflow_tensor_t flexflow_model_add_reshape(flexflow_model_t, flexflow_tensor_t input, int n, const int* shape, const char* name);
flexflow_tensor_t flexflow_model_add_batch_matmul(flexflow_model_t, flexflow_tensor_t a, flexflow_tensor_t b, int a_seq_length_dim, int b_seq_length_dim);
flexflow_initializer_t flexflow_zero_initializer_create(void);
flexflow_initializer_t flexflow_uniform_initializer_create(int seed, float min, float max);
flexflow_initializer_t flexflow_norm_initializer_create(int seed, float mean, float stddev);
flexflow_initializer_t flexflow_glorot_uniform_initializer_create(int seed);
flexflow_sgd_optimizer_t flexflow_sgd_optimizer_create(flexflow_model_t, double lr, double momentum, bool nesterov, double weight_decay);
void flexflow_model_set_sgd_optimizer(flexflow_model_t, flexflow_sgd_optimizer_t);
/* [ref: python/flexflow_c.h:398-400,572-588] */
flexflow_adam_optimizer_t flexflow_adam_optimizer_create(flexflow_model_t, double alpha, double beta1, double beta2, double weight_decay, double epsilon);
void flexflow_adam_optimizer_set_lr(flexflow_adam_optimizer_t, double lr);
void flexflow_model_set_adam_optimizer(flexflow_model_t, flexflow_adam_optimizer_t);
void flexflow_model_compile(flexflow_model_t, int loss_type, const int* metrics, int nb_metrics, int comp_mode);
void flexflow_model_init_layers(flexflow_model_t);
void flexflow_model_reset_metrics(flexflow_model_t);
void flexflow_model_forward(flexflow_model_t, int seq_length);
void flexflow_model_zero_gradients(flexflow_model_t);
void flexflow_model_backward(flexflow_model_t, int seq_length);
void flexflow_model_update(flexflow_model_t);
void flexflow_model_begin_trace(flexflow_model_t, int trace_id);
void flexflow_model_end_trace(flexflow_model_t, int trace_id);
void flexflow_model_sync(flexflow_model_t);
void flexflow_model_get_perf_metrics(flexflow_model_t, flexflow_perf_metrics_t* out);
flexflow_tensor_t flexflow_model_get_label_tensor(flexflow_model_t);
int  flexflow_model_get_num_layers(flexflow_model_t);
const char* flexflow_model_get_layer_name(flexflow_model_t, int layer);
int  flexflow_model_get_layer_num_weights(flexflow_model_t, int layer);
flexflow_tensor_t flexflow_model_get_parameter(flexflow_model_t, int layer, int index);   /* 0 kernel, 1 bias */
flexflow_tensor_t flexflow_model_get_layer_output(flexflow_model_t, int layer);
void* flexflow_model_get_stream(flexflow_model_t);
int  flexflow_model_uses_graph(flexflow_model_t);
const char* flexflow_model_get_backend_name(flexflow_model_t);   /* ffh_backend_name() of the kernel library the model loaded: "hip-gfx950" | "oracle-cpu" */
const char* flexflow_model_get_backend_path(flexflow_model_t);   /* ... and the file it was loaded from */
void flexflow_model_set_trace_mode(flexflow_model_t, int mode);   /* 0: replay a trace only where that is not slower than launching it (decided on its first calls); 1: always replay */
int  flexflow_model_trace_replays(flexflow_model_t, int trace_id);   /* 0 once the adaptive mode has settled on eager launches for this trace */
int64_t flexflow_model_get_counter(flexflow_model_t, const char* name);   /* diagnostics for tests: "mlp_chain_fwd_calls", "mlp_chain_bwd_calls"; -1: unknown */

/* Tensor / Parameter host<->device [ref: flexflow_parameter_set_weights_float, python/flexflow_c.h:498-546] */
int  flexflow_tensor_get_num_dims(flexflow_tensor_t);
void flexflow_tensor_get_dims(flexflow_tensor_t, int* dims);            /* natural order: dims[0] = batch */
int64_t flexflow_tensor_get_local_rows(flexflow_tensor_t);
bool flexflow_tensor_is_local(flexflow_tensor_t);                       /* false: table owned by another rank */
void* flexflow_tensor_get_device_ptr(flexflow_tensor_t);                /* address of element (0, 0) in the backend's memory (tests / tools: on-device
                                                                           comparisons of tables too large to copy out); NULL when not local */
int64_t flexflow_tensor_get_ld(flexflow_tensor_t);                      /* elements between consecutive rows */
void flexflow_tensor_set_float(flexflow_tensor_t, flexflow_model_t, const int* dims, int num_dims, const float* data);
void flexflow_tensor_set_int64(flexflow_tensor_t, flexflow_model_t, const int* dims, int num_dims, const int64_t* data);
void flexflow_tensor_get_float(flexflow_tensor_t, flexflow_model_t, float* data);
void flexflow_tensor_get_int64(flexflow_tensor_t, flexflow_model_t, int64_t* data);
void flexflow_tensor_get_grad_float(flexflow_tensor_t, flexflow_model_t, float* data);

/* DLRM application (examples/cpp/DLRM) */
flexflow_dlrm_t flexflow_dlrm_create(int argc, char** argv, const ffcomm* comm);
void flexflow_dlrm_destroy(flexflow_dlrm_t);
flexflow_model_t flexflow_dlrm_get_model(flexflow_dlrm_t);
int  flexflow_dlrm_get_num_samples(flexflow_dlrm_t);
int  flexflow_dlrm_get_num_tables(flexflow_dlrm_t);
flexflow_tensor_t flexflow_dlrm_get_sparse_input(flexflow_dlrm_t, int table);
flexflow_tensor_t flexflow_dlrm_get_dense_input(flexflow_dlrm_t);
void flexflow_dlrm_warmup(flexflow_dlrm_t);
void flexflow_dlrm_train_steps(flexflow_dlrm_t, int steps, bool trace);
double flexflow_dlrm_run_epochs(flexflow_dlrm_t);
/* average device time (ms) of `iters` back-to-back launches, HIP events on the launch stream:
 * which = 0 embedding gather (all owned tables, one launch), 1 fused embedding backward + SGD,
 *         2 whole training step (forward, zero_gradients, backward, update; traced if enabled) */
void flexflow_dlrm_probe_step(flexflow_dlrm_t, int iters, float* out_ms, int nout);   /* in-step event intervals, see ffmodel_c.cc; collective at world_size > 1 */
float flexflow_dlrm_time_kernel(flexflow_dlrm_t, int which, int iters);

#ifdef __cplusplus
}
#endif
#endif
