"""Special-token strings (reference valley/util/config.py:1-13)."""
from valley_amd.valley_model import (DEFAULT_BOS_TOKEN, DEFAULT_EOS_TOKEN, DEFAULT_IM_END_TOKEN,  # noqa: F401
                                     DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_TOKEN,
                                     DEFAULT_PAD_TOKEN, DEFAULT_UNK_TOKEN, DEFAULT_VI_END_TOKEN,
                                     DEFAULT_VI_START_TOKEN, DEFAULT_VIDEO_FRAME_TOKEN, DEFAULT_VIDEO_TOKEN,
                                     IGNORE_INDEX)
