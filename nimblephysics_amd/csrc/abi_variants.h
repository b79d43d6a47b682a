/* abi_variants.h — the library is built from ONE set of sources in four instantiations of the contact stage:
 *   NBL_MAXC = 8   24 LCP rows, 16 colliders, 32 collider pairs   (suffix _c8,  device namespace nbl)
 *   NBL_MAXC = 16  48 LCP rows, 32 colliders, 64 collider pairs   (suffix _c16, device namespace nbl_c16)
 *   NBL_MAXC = 64  192 LCP rows, 64 colliders, 512 collider pairs (suffix _c64, device namespace nbl_c64): the GENERAL instantiation, whose
 *                  dense contact kernels loop over the rows (gen_contact.hip) instead of mapping them to lanes (coop_kernels.hip)
 *   NBL_MAXC = 128 the same general code with 384 rows (suffix _c128, device namespace nbl_c128): only for models that ask for more than 64
 *                  contact slots (its per-world scratch and record are four times the 192-row build's)
 * Each instantiation is one translation unit (nimble_amd.hip compiled with -DNBL_MAXC=.. -DNBL_VARIANT_SUFFIX=..); this header,
 * included before include/nimble_amd.h, renames the ABI's entry points and its opaque handle type with the suffix so that both fit in
 * one shared library.  nimble_amd_dispatch.cpp exports the names of include/nimble_amd.h and hands every model to the instantiation
 * that fits it (nbl_model_create).  Without NBL_VARIANT_SUFFIX nothing is renamed: nimble_amd.hip alone is then the whole library
 * in its 8-contact form (the developer builds of tools/ compile it that way).
 */
#ifndef NBL_ABI_VARIANTS_H
#define NBL_ABI_VARIANTS_H

/* internal status of nbl_model_create: the model exceeds THIS instantiation's collider / pair / contact budget (the dispatcher then
 * tries the larger one; callers of the library see NBL_E_UNSUPPORTED) */
#define NBL_E_CAPACITY -100

#ifdef NBL_VARIANT_SUFFIX
#define NBL_CAT2_(a, b) a##b
#define NBL_CAT2(a, b) NBL_CAT2_(a, b)
#define NBL_V(name) NBL_CAT2(name, NBL_VARIANT_SUFFIX)

#define nbl_model NBL_V(nbl_model)
#define nbl_backward_inertia NBL_V(nbl_backward_inertia)
#define nbl_device_count NBL_V(nbl_device_count)
#define nbl_get_timing NBL_V(nbl_get_timing)
#define nbl_kernel_count NBL_V(nbl_kernel_count)
#define nbl_kernel_name NBL_V(nbl_kernel_name)
#define nbl_kernel_timing NBL_V(nbl_kernel_timing)
#define nbl_last_error NBL_V(nbl_last_error)
#define nbl_model_create NBL_V(nbl_model_create)
#define nbl_model_destroy NBL_V(nbl_model_destroy)
#define nbl_model_lcp_rows NBL_V(nbl_model_lcp_rows)
#define nbl_model_num_action NBL_V(nbl_model_num_action)
#define nbl_model_num_dofs NBL_V(nbl_model_num_dofs)
#define nbl_num_inertia_params NBL_V(nbl_num_inertia_params)
#define nbl_rollout_backward NBL_V(nbl_rollout_backward)
#define nbl_rollout_backward_checkpointed NBL_V(nbl_rollout_backward_checkpointed)
#define nbl_rollout_backward_inertia NBL_V(nbl_rollout_backward_inertia)
#define nbl_rollout_checkpoint_bytes NBL_V(nbl_rollout_checkpoint_bytes)
#define nbl_rollout_forward NBL_V(nbl_rollout_forward)
#define nbl_rollout_forward_checkpointed NBL_V(nbl_rollout_forward_checkpointed)
#define nbl_rollout_workspace_bytes NBL_V(nbl_rollout_workspace_bytes)
#define nbl_saved_bytes NBL_V(nbl_saved_bytes)
#define nbl_selftest_lcp_dantzig NBL_V(nbl_selftest_lcp_dantzig)
#define nbl_selftest_lcp_dantzig_timed NBL_V(nbl_selftest_lcp_dantzig_timed)
#define nbl_selftest_lcp_cascade NBL_V(nbl_selftest_lcp_cascade)
#define nbl_selftest_pinv NBL_V(nbl_selftest_pinv)
#define nbl_selftest_pinv_rows NBL_V(nbl_selftest_pinv_rows)
#define nbl_set_body_inertia NBL_V(nbl_set_body_inertia)
#define nbl_set_body_inertias NBL_V(nbl_set_body_inertias)
#define nbl_set_inertia_params NBL_V(nbl_set_inertia_params)
#define nbl_set_inertia_params_on NBL_V(nbl_set_inertia_params_on)
#define nbl_set_launch_lanes NBL_V(nbl_set_launch_lanes)
#define nbl_set_slices NBL_V(nbl_set_slices)
#define nbl_set_timing NBL_V(nbl_set_timing)
#define nbl_slices_for NBL_V(nbl_slices_for)
#define nbl_set_deferred_join NBL_V(nbl_set_deferred_join)
#define nbl_slice_stream NBL_V(nbl_slice_stream)
#define nbl_join_slices NBL_V(nbl_join_slices)
#define nbl_fork_slices NBL_V(nbl_fork_slices)
#define nbl_step_backward NBL_V(nbl_step_backward)
#define nbl_step_forward NBL_V(nbl_step_forward)
#define nbl_transpose_from_soa NBL_V(nbl_transpose_from_soa)
#define nbl_transpose_to_soa NBL_V(nbl_transpose_to_soa)
#define nbl_version NBL_V(nbl_version)
#define nbl_workspace_bytes NBL_V(nbl_workspace_bytes)
#define nbl_model_max_contacts NBL_V(nbl_model_max_contacts)
#elif !defined(NBL_DISPATCHER)
/* the stand-alone 8-contact build has nobody to hand the model on to */
#undef NBL_E_CAPACITY
#define NBL_E_CAPACITY NBL_E_UNSUPPORTED
#endif /* NBL_VARIANT_SUFFIX */

#endif /* NBL_ABI_VARIANTS_H */
