"""Our own configuration files for the two test cases the path is pinned to: the namelist dictionaries the reference's Python
harness hands to the model, key for key (values of exp/test_cases/held_suarez/held_suarez_test_case.py:45-98 and
exp/test_cases/frierson/frierson_test_case.py:49-170), including the framework groups this implementation ignores
(diag_manager_nml, fms_nml, fms_io_nml, betts_miller_nml).  Use with isca_amd.atmosphere.atmosphere_init or Experiment.update_namelist;
the resolution comes from Experiment.set_resolution / the `resolution` argument like in the reference scripts.
"""
import copy

HELD_SUAREZ = {
    "main_nml": {"dt_atmos": 600, "days": 30, "calendar": "thirty_day", "current_date": [2000, 1, 1, 0, 0, 0]},
    "atmosphere_nml": {"idealized_moist_model": False},
    "spectral_dynamics_nml": {"damping_order": 4, "water_correction_limit": 200.e2, "reference_sea_level_press": 1.0e5,
                              "valid_range_t": [100., 800.], "initial_sphum": 0.0, "vert_coord_option": "uneven_sigma",
                              "scale_heights": 6.0, "exponent": 7.5, "surf_res": 0.5},
    "hs_forcing_nml": {"t_zero": 315., "t_strat": 200., "delh": 60., "delv": 10., "eps": 0., "sigma_b": 0.7, "ka": -40., "ks": -4.,
                       "kf": -1., "do_conserve_energy": True},
    "diag_manager_nml": {"mix_snapshot_average_fields": False},
    "fms_nml": {"domains_stack_size": 600000},
    "fms_io_nml": {"threading_write": "single", "fileset_write": "single"},
}

FRIERSON = {
    "main_nml": {"days": 30, "hours": 0, "minutes": 0, "seconds": 0, "dt_atmos": 720, "current_date": [1, 1, 1, 0, 0, 0],
                 "calendar": "thirty_day"},
    "idealized_moist_phys_nml": {"do_damping": True, "turb": True, "mixed_layer_bc": True, "do_virtual": False, "do_simple": True,
                                 "roughness_mom": 3.21e-05, "roughness_heat": 3.21e-05, "roughness_moist": 3.21e-05,
                                 "two_stream_gray": True, "convection_scheme": "SIMPLE_BETTS_MILLER"},
    "vert_turb_driver_nml": {"do_mellor_yamada": False, "do_diffusivity": True, "do_simple": True, "constant_gust": 0.0, "use_tau": False},
    "diffusivity_nml": {"do_entrain": False, "do_simple": True},
    "surface_flux_nml": {"use_virtual_temp": False, "do_simple": True, "old_dtaudv": True},
    "atmosphere_nml": {"idealized_moist_model": True},
    "mixed_layer_nml": {"tconst": 285., "prescribe_initial_dist": True, "evaporation": True, "depth": 2.5, "albedo_value": 0.31},
    "qe_moist_convection_nml": {"rhbm": 0.7, "Tmin": 160., "Tmax": 350.},
    "betts_miller_nml": {"rhbm": .7, "do_simp": False, "do_shallower": True},
    "lscale_cond_nml": {"do_simple": True, "do_evap": True},
    "sat_vapor_pres_nml": {"do_simple": True},
    "damping_driver_nml": {"do_rayleigh": True, "trayfric": -0.25, "sponge_pbottom": 5000., "do_conserve_energy": True},
    "two_stream_gray_rad_nml": {"rad_scheme": "frierson", "do_seasonal": False, "atm_abs": 0.2},
    "diag_manager_nml": {"mix_snapshot_average_fields": False},
    "fms_nml": {"domains_stack_size": 600000},
    "fms_io_nml": {"threading_write": "single", "fileset_write": "single"},
    "spectral_dynamics_nml": {"damping_order": 4, "water_correction_limit": 200.e2, "reference_sea_level_press": 1.0e5, "num_levels": 25,
                              "valid_range_t": [100., 800.], "initial_sphum": [2.e-6], "vert_coord_option": "input", "surf_res": 0.5,
                              "scale_heights": 11.0, "exponent": 7.0, "robert_coeff": 0.03},
    "vert_coordinate_nml": {
        "bk": [0.000000, 0.0117665, 0.0196679, 0.0315244, 0.0485411, 0.0719344, 0.1027829, 0.1418581, 0.1894648, 0.2453219, 0.3085103,
               0.3775033, 0.4502789, 0.5244989, 0.5977253, 0.6676441, 0.7322627, 0.7900587, 0.8400683, 0.8819111, 0.9157609, 0.9422770,
               0.9625127, 0.9778177, 0.9897489, 1.0000000],
        "pk": [0.0] * 26},
}


def held_suarez():
    return copy.deepcopy(HELD_SUAREZ)


def frierson():
    return copy.deepcopy(FRIERSON)
