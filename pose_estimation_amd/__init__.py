"""MI355X-native RANSAC pose-hypothesis path (P3P / P4Pf solve + reprojection scoring)."""
from ._lib import PoseAmdError, LIB_PATH  # noqa: F401
from .api import (Context, P3p, P4pf, Query, QueryProcessor, QueryProcessorBasic, QueryProcessorAdvanced,  # noqa: F401
                  SELECT_INLIERS, SELECT_COVERAGE, SAMPLE_BASIC, SAMPLE_ADVANCED, default_context)
