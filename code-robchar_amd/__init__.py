"""robchar_amd - MI355X-native Monte-Carlo robustness characterisation (RobChar hot path).

Drop-in surface (mirrors of the reference's modules for the MC path only):
    noise_model.structured_perturbation / noise_function      (reference noise_model.py)
    mcsim.MCDataSim                                           (reference mcsim.py:200-510)
    metrics.wd_from_ideal / RIM_p / compute_dkw_error / ...   (reference wd_sortof_fast_implementation.py)
    naming.ExperimentNamer                                    (reference noise_analysis.py:33-49)
Compute goes through librobchar_hip.so (include/robchar_hip.h) - see backend.py.
"""
__version__ = "0.1.0"
