"""MI355X-native replacement for the Warp-backed hot path of ppr-diffphys.

Only the path of SURVEY.md section 8 lives here: the model compiler
(:mod:`sim`, :mod:`import_urdf`, :mod:`robots`), the ctypes binding of the HIP
library (:mod:`hip_backend`) and the two autograd boundaries of the reference
(:mod:`dp_model`: ``ForwardWarp`` / ``ForwardKinematics``).
"""
__version__ = "0.1.0"
