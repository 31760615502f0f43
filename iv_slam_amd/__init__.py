"""iv_slam_amd -- MI355X-native visual front end of IV-SLAM (ORB extract + stereo/Hamming match + introspection).

Only what the hot path needs: csrc/ (HIP kernels + the C-ABI, built into libivfront.so) and the host-side
mirror of the reference's ORBextractor / ORBmatcher interfaces.  No CPU fallback.
"""
from ._lib import KP_DTYPE, IvfError, load  # noqa: F401
from .orb import (ORBextractor, ORBmatcher, ORBVocabulary, ComputeStereoMatches, GetFeaturesInArea,  # noqa: F401
                  ComputeDistinctiveDescriptors, DeviceFrame)
from .frontend import StereoFrontend  # noqa: F401
from .fcn import IntrospectionFCN  # noqa: F401
from .rectify import initUndistortRectifyMap, Remap  # noqa: F401
from .track import BatchTracker  # noqa: F401
