/* TEST INFRASTRUCTURE.  The drop-in boundary of this repo is a C ABI (include/pfn_hip.h): this program is a C99 client of it -- it includes the header as C,
 * dlopens libpfn_hip.so, resolves every entry point named on its command line (tests/test_host.py passes the names it finds in the header) and calls the
 * host-side ones, which need no GPU: version, parameter packing, workspace sizing, the top-layer row rule, the error path.
 *     gcc -std=c99 -Iinclude tests/cabi_check.c -ldl -o cabi_check && ./cabi_check <libpfn_hip.so> pfn_abi_version pfn_param_count ... */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "pfn_hip.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "cabi_check: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: cabi_check <library> <symbol> ...\n"); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "cabi_check: dlopen: %s\n", dlerror()); return 1; }
  for (int i = 2; i < argc; ++i)
    if (!dlsym(lib, argv[i])) { fprintf(stderr, "cabi_check: %s is declared in pfn_hip.h but not exported\n", argv[i]); return 1; }

  int (*abi_version)(void) = (int (*)(void))dlsym(lib, "pfn_abi_version");
  int64_t (*param_count)(const pfn_model_desc*) = (int64_t (*)(const pfn_model_desc*))dlsym(lib, "pfn_param_count");
  int (*param_layout)(const pfn_model_desc*, int64_t*, int64_t*, int) = (int (*)(const pfn_model_desc*, int64_t*, int64_t*, int))dlsym(lib, "pfn_param_layout");
  int64_t (*shadow_bytes)(const pfn_model_desc*) = (int64_t (*)(const pfn_model_desc*))dlsym(lib, "pfn_shadow_bytes");
  int64_t (*workspace_bytes)(const pfn_model_desc*, int, int) = (int64_t (*)(const pfn_model_desc*, int, int))dlsym(lib, "pfn_workspace_bytes");
  int64_t (*top_rows)(const pfn_model_desc*, int, int, int, int) = (int64_t (*)(const pfn_model_desc*, int, int, int, int))dlsym(lib, "pfn_top_layer_rows");
  const char* (*last_error)(void) = (const char* (*)(void))dlsym(lib, "pfn_last_error_string");
  int (*set_tuning)(int, int) = (int (*)(int, int))dlsym(lib, "pfn_set_tuning");

  CHECK(abi_version() == PFN_ABI_VERSION);
  /* BASELINE configs[1]: 18 features, emsize 512, 4 heads, nhid 1024, 6 layers, 1000 bars */
  pfn_model_desc d;
  memset(&d, 0, sizeof(d));
  d.num_features = 18; d.emsize = 512; d.nhead = 4; d.nhid = 1024; d.nlayers = 6; d.n_out = 1000;
  d.precision = PFN_PREC_BF16; d.ln_eps = 1e-5f; d.dropout = 0.f;
  const int n = param_layout(&d, 0, 0, 0);
  CHECK(n == 4 + 12 * 6 + 4);
  int64_t offs[128], nums[128], total = 0;
  CHECK(param_layout(&d, offs, nums, 128) == n);
  for (int i = 0; i < n; ++i) total += nums[i];
  CHECK(total == 14177768);                       /* parameters of the north-star model */
  CHECK(param_count(&d) >= offs[n - 1] + nums[n - 1]);
  CHECK(shadow_bytes(&d) > 2 * total);            /* operand-precision copy + pre-transposed 2-D weights */
  CHECK(workspace_bytes(&d, 32, 2000) > 0 && workspace_bytes(&d, 64, 2000) > workspace_bytes(&d, 32, 2000));
  CHECK(top_rows(&d, 32, 2000, 1604, 0) == (int64_t)(2000 - 1604) * 32);
  CHECK(top_rows(&d, 32, 2000, 100, 0) == (int64_t)2000 * 32);
  /* the schedule is part of the descriptor (ABI 6): the same bits reach a forward and the backward that reads its workspace */
  d.schedule = PFN_SCHED_TOP_LAYER_ALL_ROWS;
  CHECK(top_rows(&d, 32, 2000, 1604, 0) == (int64_t)2000 * 32);
  d.schedule = 0;
  /* the test / profiling knob only changes the default for NEW descriptors */
  int (*default_schedule)(void) = (int (*)(void))dlsym(lib, "pfn_default_schedule");
  CHECK(default_schedule() == 0);
  CHECK(set_tuning(PFN_TUNE_TOP_LAYER_TEST_ROWS, 0) == PFN_OK && default_schedule() == PFN_SCHED_TOP_LAYER_ALL_ROWS);
  CHECK(top_rows(&d, 32, 2000, 1604, 0) == (int64_t)(2000 - 1604) * 32);
  CHECK(set_tuning(PFN_TUNE_TOP_LAYER_TEST_ROWS, 1) == PFN_OK && default_schedule() == 0);
  CHECK(set_tuning(12345, 0) < 0);
  /* error path: head dim 100 (the reference's train() default emsize 200 / nhead 2) is refused with a message, not computed wrongly */
  d.emsize = 200; d.nhead = 2; d.nhid = 200;
  CHECK(param_count(&d) < 0 && strstr(last_error(), "head dim") != 0);
  /* ABI 7: the GP sampler is told how large the caller's K_ws is -- an allocation below B*S*S*4 bytes is refused before anything touches the device, and
   * pfn_gp_workspace_bytes is what a caller should allocate (the matrix + the trailing update's plane scratch) */
  int64_t (*gp_ws)(int, int) = (int64_t (*)(int, int))dlsym(lib, "pfn_gp_workspace_bytes");
  int (*gp_sample)(float*, float*, float*, float*, int64_t, const float*, const float*, const float*, int, int, int, int, int, int, uint64_t, uint64_t, int32_t*, void*) =
      (int (*)(float*, float*, float*, float*, int64_t, const float*, const float*, const float*, int, int, int, int, int, int, uint64_t, uint64_t, int32_t*, void*))dlsym(lib, "pfn_gp_prior_sample");
  CHECK(gp_ws(2, 2000) >= (int64_t)2 * 2000 * 2000 * 4 && gp_ws(2, 2000) < (int64_t)2 * 2000 * 2000 * 6);
  {
    float dummy[4]; int32_t info[2];
    CHECK(gp_sample(dummy, dummy, dummy, dummy, (int64_t)2 * 2000 * 2000 * 4 - 1, dummy, dummy, dummy, 2, 2000, 18, 0, 1, 1, 0, 0, info, 0) == PFN_ERR_ARGUMENT);
    CHECK(strstr(last_error(), "K_ws_bytes") != 0);
  }
  /* the deterministic schedule is a descriptor bit like the others and asks for its scratch in the workspace */
  d.emsize = 512; d.nhead = 4; d.nhid = 1024;
  {
    const int64_t plain = workspace_bytes(&d, 8, 2000);
    d.schedule = PFN_SCHED_DETERMINISTIC;
    CHECK(workspace_bytes(&d, 8, 2000) > plain);
    d.schedule = 1 << 20;                          /* an unknown bit is refused */
    CHECK(workspace_bytes(&d, 8, 2000) < 0 && strstr(last_error(), "schedule") != 0);
    d.schedule = 0;
    /* ABI 8: fp16 operands -- same workspace as bf16 plus the per-dataset key shift (keys centred by default with fp16; PFN_SCHED_NO_KEY_CENTERING drops it,
     * PFN_SCHED_KEY_CENTERING adds it to bf16) */
    d.precision = PFN_PREC_FP16;
    CHECK(workspace_bytes(&d, 8, 2000) > plain && shadow_bytes(&d) > 0);
    {
      const int64_t centred = workspace_bytes(&d, 8, 2000);
      d.schedule = PFN_SCHED_NO_KEY_CENTERING;
      CHECK(workspace_bytes(&d, 8, 2000) == plain);
      d.precision = PFN_PREC_BF16; d.schedule = PFN_SCHED_KEY_CENTERING;
      CHECK(workspace_bytes(&d, 8, 2000) == centred);
      d.schedule = 0;
    }
  }
  /* ragged batches (ABI 7): the arguments are checked before anything is launched -- missing per-dataset arrays or an impossible row count are refused */
  {
    int (*fwd_ragged)(const pfn_model_desc*, const float*, const void*, const float*, int64_t, int64_t, const float*, int64_t, int64_t, int, int,
                      const int32_t*, const int64_t*, int, int, int64_t, void*, int64_t, float*, void*, int, uint64_t) =
        (int (*)(const pfn_model_desc*, const float*, const void*, const float*, int64_t, int64_t, const float*, int64_t, int64_t, int, int,
                 const int32_t*, const int64_t*, int, int, int64_t, void*, int64_t, float*, void*, int, uint64_t))dlsym(lib, "pfn_stack_forward_ragged");
    float dummy[4]; int32_t seps[2] = {3, 5}; int64_t offs[3] = {0, 5, 8};
    CHECK(fwd_ragged(&d, dummy, dummy, dummy, 36, 18, dummy, 2, 1, 2, 8, 0, offs, 3, 5, 8, dummy, 1 << 20, dummy, 0, 0, 0) == PFN_ERR_ARGUMENT);      /* no sep_of */
    CHECK(fwd_ragged(&d, dummy, dummy, dummy, 36, 18, dummy, 2, 1, 2, 8, seps, offs, 3, 5, 17, dummy, 1 << 20, dummy, 0, 0, 0) == PFN_ERR_ARGUMENT);   /* more test rows than rows */
    CHECK(fwd_ragged(&d, dummy, dummy, dummy, 36, 18, dummy, 2, 1, 2, 8, seps, offs, 6, 5, 8, dummy, 1 << 20, dummy, 0, 0, 0) == PFN_ERR_ARGUMENT);    /* sep_min > sep_max */
  }
  printf("cabi_check ok: ABI %d, %d parameter tensors, %lld parameters\n", abi_version(), n, (long long)total);
  dlclose(lib);
  return 0;
}
