/* A plain C99 host of the C ABI (include/styler_hip.h): loads libstyler_hip.so, checks the ABI version and the
 * argument-validation contract (negative STYLER_E* codes, returned before any HIP call, so this runs without a GPU).
 * Built and run by tests/test_host_cpu.py::test_c_host_links_and_validates_arguments. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdint.h>

#include "styler_hip.h"

typedef int (*abi_version_fn)(void);
typedef int (*leaky_sum_fn)(const float*, const float*, const float*, float*, int64_t, float, float, void*);
typedef int (*conv_gemm_pad_fn)(const float*, int64_t, const void*, const float*, const float*, const float*, int64_t,
                                float*, int64_t, int, int, int, int, int, int, int, int, void*);
typedef int (*pack_plan_fn)(const int64_t*, int, int, int32_t*, int32_t*, int32_t*, int64_t*, void*);

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  abi_version_fn ver = (abi_version_fn)dlsym(h, "styler_abi_version");
  leaky_sum_fn leaky = (leaky_sum_fn)dlsym(h, "styler_leaky_sum");
  conv_gemm_pad_fn gemm = (conv_gemm_pad_fn)dlsym(h, "styler_conv_gemm_pad");
  pack_plan_fn plan = (pack_plan_fn)dlsym(h, "styler_pack_plan");
  if (!ver || !leaky || !gemm || !plan) return 4;
  float buf[8] = {0};
  int rc_null = leaky(NULL, NULL, NULL, buf, 8, 1.0f, 0.1f, NULL);                 /* missing input  -> EINVAL */
  int rc_align = leaky(buf + 1, NULL, NULL, buf, 4, 1.0f, 0.1f, NULL);             /* unaligned ptr  -> EALIGN */
  int rc_kw = gemm(buf, 4, buf, NULL, NULL, NULL, 0, buf, 4, 1, 1, 4, 4, 11, 5, STYLER_ACT_NONE, STYLER_PREC_F32,
                   NULL);                                                          /* kw > 9         -> EINVAL */
  int rc_plan = plan(NULL, 1, 1, NULL, NULL, NULL, NULL, NULL);                    /* null tables    -> EINVAL */
  printf("abi=%d null=%d align=%d kw=%d plan=%d\n", ver(), rc_null, rc_align, rc_kw, rc_plan);
  return (ver() == 1 && rc_null == STYLER_EINVAL && rc_align == STYLER_EALIGN && rc_kw == STYLER_EINVAL &&
          rc_plan == STYLER_EINVAL) ? 0 : 1;
}
