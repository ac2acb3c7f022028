"""MI355X-native hot path of the Explicit-Context-Mapping stereo network.

The directory name is not a Python identifier; import it through the root shim:
    import ecm_amd as ecm;  model = ecm.get_model("cmfsm").cuda()
"""
from . import _lib, ops                                                    # noqa: F401
from .models import (cmfsm, convbn_3d, disparityregression, eight_related_context_mapping,   # noqa: F401
                     feature_extraction, get_model, hourglass, matchshifted, similarity_measure1)

__all__ = ["get_model", "cmfsm", "hourglass", "convbn_3d", "disparityregression", "matchshifted",
           "eight_related_context_mapping", "similarity_measure1", "feature_extraction", "ops"]
