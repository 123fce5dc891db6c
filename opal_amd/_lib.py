"""ctypes loader for libopal_amd.so (the C ABI declared in include/opal_amd.h).

There is no CPU fallback: if the shared library is missing or does not load,
importing the product path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPAL_AMD_LIB") or os.path.join(_HERE, "lib", "libopal_amd.so")

FN1 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
FN2 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_void_p)
FN3 = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
EXCHANGE2_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double),
                        C.POINTER(C.c_size_t), C.c_int)

vp = C.c_void_p
dbl = C.c_double
sz = C.c_size_t
pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/opal_amd.h one to one
SIGNATURES = {
    "opal_last_error": (C.c_char_p, []),
    "opal_version": (C.c_char_p, []),
    "opal_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "opal_set_device": (C.c_int, [C.c_int]),
    "opal_device_identity": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "opal_design_unbalanced": (C.c_int, [C.c_int, sz, dbl, sz, vp, vp]),
    "opal_design_balanced": (C.c_int, [C.c_int, sz, dbl, dbl, sz, FN1, vp, vp, vp]),
    "opal_grid_build": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, C.c_int, dbl, dbl, vp, vp, vp]),
    "opal_grid_free": (C.c_int, [vp]),
    "opal_grid_rank": (C.c_int, [vp]),
    "opal_grid_ngrids": (C.c_int, [vp]),
    "opal_grid_to_left": (C.c_int, [vp]),
    "opal_grid_to_right": (C.c_int, [vp]),
    "opal_grid_xmin": (dbl, [vp]),
    "opal_grid_dx": (dbl, [vp]),
    "opal_grid_size": (sz, [vp]),
    "opal_grid_len": (sz, [vp]),
    "opal_grid_left_bdy_size": (sz, [vp]),
    "opal_grid_right_bdy_size": (sz, [vp]),
    "opal_grid_min_size": (sz, []),
    "opal_grid_clear": (C.c_int, [vp]),
    "opal_grid_deposit": (C.c_int, [vp, vp, dbl]),
    "opal_grid_synchronize": (C.c_int, [vp, vp, FN2, FN2, vp, dbl, dbl]),
    "opal_grid_advance": (C.c_int, [vp, dbl]),
    "opal_grid_fields_at": (C.c_int, [vp, sz, vp, vp, vp, vp]),
    "opal_grid_em_field_energy": (C.c_int, [vp, C.POINTER(dbl)]),
    "opal_grid_interpolate": (C.c_int, [vp, vp]),
    "opal_grid_write_data": (C.c_int, [vp, C.c_char_p, sz]),
    "opal_grid_get_raw": (C.c_int, [vp, C.c_int, vp]),
    "opal_grid_set_raw": (C.c_int, [vp, C.c_int, vp]),
    "opal_grid_assume_electrostatic": (C.c_int, [vp]),
    "opal_pop_create": (C.c_int, [pp, C.c_int, sz, vp]),
    "opal_pop_free": (C.c_int, [vp]),
    "opal_pop_new": (C.c_int, [pp, C.c_int, sz, FN1, FN3, FN3, FN3, vp, vp, C.c_uint64, dbl, vp]),
    "opal_loader_generate": (C.c_int, [C.c_int, sz, FN1, FN3, FN3, FN3, vp, dbl, dbl, sz, C.c_uint64, sz, C.POINTER(sz), vp, vp, vp, vp, vp, vp]),
    "opal_format": (C.c_int, [C.c_int, dbl, C.c_char_p, sz]),
    "opal_pop_set_charge_state": (C.c_int, [vp, dbl, dbl, dbl]),
    "opal_pop_upload_soa": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, dbl, dbl]),
    "opal_pop_download_soa": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "opal_pop_len": (C.c_int, [vp, C.POINTER(sz)]),
    "opal_pop_live": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "opal_pop_advance": (C.c_int, [vp, vp, vp, dbl]),
    "opal_pop_advance_deposit": (C.c_int, [vp, vp, vp, dbl]),
    "opal_pop_total_kinetic_energy": (C.c_int, [vp, C.POINTER(dbl)]),
    "opal_pop_sort": (C.c_int, [vp, vp]),
    "opal_pop_set_sort_interval": (C.c_int, [vp, C.c_int]),
    "opal_pop_counters": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "opal_pop_debug_misses": (C.c_int, [vp, C.c_int, vp]),
    "opal_set_elision": (C.c_int, [C.c_int]),
    "opal_pop_track_extras": (C.c_int, [vp, C.c_int]),
    "opal_pop_track_fields": (C.c_int, [vp, C.c_uint]),
    "opal_pop_keep_prev_x": (C.c_int, [vp, C.c_int]),
    "opal_pop_download_extras": (C.c_int, [vp, vp, vp, vp, vp]),
    "opal_pop_download_emission_state": (C.c_int, [vp, vp, vp]),
    "opal_pop_upload_emission_state": (C.c_int, [vp, sz, vp, vp]),
    "opal_emit_radiation": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, vp]),
    "opal_absorb": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, dbl, dbl, dbl, dbl, C.POINTER(sz), C.POINTER(sz)]),
    "opal_pop_download_absorption_state": (C.c_int, [vp, vp, vp, vp]),
    "opal_pop_upload_absorption_state": (C.c_int, [vp, sz, vp, vp, vp]),
    "opal_pop_upload_chi": (C.c_int, [vp, sz, vp]),
    "opal_pop_upload_yz": (C.c_int, [vp, sz, vp, vp]),
    "opal_qed_airy_batch": (C.c_int, [sz, vp, vp]),
    "opal_qed_cross_sections_batch": (C.c_int, [sz, vp, vp, vp, vp, vp, vp]),
    "opal_qed_sample_batch": (C.c_int, [sz, vp, vp, vp, vp, vp, C.c_int, vp]),
    "opal_qed_randoms_batch": (C.c_int, [sz, C.c_uint64, vp, C.c_uint32, vp]),
    "opal_qed_radiate_batch": (C.c_int, [sz, vp, vp, vp, vp, vp, vp, vp, vp]),
    "opal_qed_rate_batch": (C.c_int, [sz, vp, vp, vp]),
    "opal_pop_hist_range": (C.c_int, [vp, vp, C.c_char_p, C.POINTER(C.c_int), vp, vp, C.POINTER(sz)]),
    "opal_pop_hist_fill": (C.c_int, [vp, vp, C.c_char_p, vp, vp, sz, pp]),
    "opal_hist_len": (sz, [vp]),
    "opal_hist_counts": (C.POINTER(dbl), [vp]),
    "opal_hist_total": (dbl, [vp]),
    "opal_hist_set_total": (C.c_int, [vp, dbl]),
    "opal_hist_dims": (C.c_int, [vp, C.POINTER(C.c_int), vp, vp, vp]),
    "opal_hist_finish": (C.c_int, [vp]),
    "opal_hist_write_fits": (C.c_int, [vp, C.c_char_p]),
    "opal_hist_free": (C.c_int, [vp]),
    "opal_hist_filename": (C.c_int, [C.c_char_p, sz, C.c_char_p, C.c_char_p, C.c_char_p, sz]),
    "opal_pop_write_data": (C.c_int, [vp, vp, C.c_char_p, sz, C.c_char_p, C.POINTER(C.c_char_p), sz]),
    "opal_link_create": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, sz, EXCHANGE_FN, vp, vp]),
    "opal_link_set_exchange2": (C.c_int, [vp, EXCHANGE2_FN]),
    "opal_link_set_bounded": (C.c_int, [vp, C.c_int]),
    "opal_link_set_growth": (C.c_int, [vp, C.c_int]),
    "opal_link_growth": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint)]),
    "opal_link_last_exchange_bytes": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz)]),
    "opal_link_rccl_ranks": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "opal_link_create_peer": (C.c_int, [pp, C.c_int, C.c_int, sz, C.c_int, C.c_int, vp]),
    "opal_link_self_test": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "opal_link_plan_selftest": (C.c_int, []),
    "opal_link_peer_export": (C.c_int, [vp, vp]),
    "opal_link_peer_connect": (C.c_int, [vp, vp, vp]),
    "opal_link_peer_status": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "opal_rccl_unique_id": (C.c_int, [C.c_char_p, vp]),
    "opal_link_create_rccl": (C.c_int, [pp, C.c_int, C.c_int, sz, C.c_int, C.c_int, C.c_char_p, vp, vp]),
    "opal_link_create_layout": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, sz, C.c_int, C.c_int, EXCHANGE_FN, vp, vp]),
    "opal_link_free": (C.c_int, [vp]),
    "opal_link_message_bytes": (sz, [vp]),
    "opal_link_halo_bytes": (sz, [vp]),
    "opal_link_particle_capacity": (sz, [vp]),
    "opal_link_set_buffers": (C.c_int, [vp, vp, vp, vp, vp]),
    "opal_link_buffer": (vp, [vp, C.c_int]),
    "opal_sim_create": (C.c_int, [pp, C.c_char_p, C.c_int, C.c_int, vp, vp]),
    "opal_sim_create_from_string": (C.c_int, [pp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, vp, vp]),
    "opal_sim_free": (C.c_int, [vp]),
    "opal_sim_step": (C.c_int, [vp, sz]),
    "opal_sim_run": (C.c_int, [vp]),
    "opal_sim_sync": (C.c_int, [vp]),
    "opal_sim_time": (dbl, [vp]),
    "opal_sim_dt": (dbl, [vp]),
    "opal_sim_total_steps": (sz, [vp]),
    "opal_sim_steps_between_outputs": (sz, [vp]),
    "opal_sim_n_outputs": (sz, [vp]),
    "opal_sim_grid": (vp, [vp]),
    "opal_sim_electrons": (vp, [vp]),
    "opal_sim_ions": (vp, [vp]),
    "opal_sim_photons": (vp, [vp]),
    "opal_sim_set_overlap": (C.c_int, [vp, C.c_int]),
    "opal_sim_overlapped_steps": (sz, [vp]),
    "opal_link_peer_wait": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(C.c_uint64), C.c_int]),
    "opal_set_loader_threads": (C.c_int, [C.c_int]),
    "opal_sim_set_recut": (C.c_int, [vp, sz, dbl, sz]),
    "opal_sim_recut_now": (C.c_int, [vp]),
    "opal_sim_drift_stats": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(sz)]),
    "opal_sim_recut_stats": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(sz), C.POINTER(dbl)]),
    "opal_recut_plan": (C.c_int, [C.c_int, vp, vp, vp, sz, sz, vp, vp, C.POINTER(dbl), C.POINTER(dbl)]),
    "opal_grid_set_self_exchange": (C.c_int, [vp, C.c_int]),
    "opal_sim_set_collectives": (C.c_int, [vp, ALLREDUCE_FN, GATHER_FN, vp]),
    "opal_grid_write_rows": (C.c_int, [vp, sz, C.c_char_p, sz]),
    "opal_sim_energies": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]),
    "opal_sim_set_fused": (C.c_int, [vp, C.c_int]),
    "opal_stream_create": (C.c_int, [pp]),
    "opal_stream_destroy": (C.c_int, [vp]),
    "opal_prof_enable": (C.c_int, [C.c_int]),
    "opal_prof_reset": (C.c_int, []),
    "opal_prof_seen": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "opal_prof_get": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(dbl)]),
    "opal_prof_forms": (C.c_int, [C.POINTER(C.c_uint64)]),
    "opal_config_from_file": (C.c_int, [pp, C.c_char_p]),
    "opal_config_from_string": (C.c_int, [pp, C.c_char_p]),
    "opal_config_free": (C.c_int, [vp]),
    "opal_config_contains": (C.c_int, [vp, C.c_char_p]),
    "opal_config_read_f64": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.POINTER(dbl)]),
    "opal_config_read_usize": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.POINTER(sz)]),
    "opal_config_read_bool": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "opal_config_eval": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(dbl), C.POINTER(dbl)]),
    "opal_config_eval_expr": (C.c_int, [vp, C.c_char_p, C.POINTER(dbl)]),
}

_lib = None


def load():
    """Load libopal_amd.so and declare every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C opal_amd/csrc). "
            "opal_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if "OPAL_AMD_LIB" in os.environ and not hasattr(lib, name):
            continue  # A/B runs of an older build (scripts/archive/ab_run2.sh); the in-tree library must export everything
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
