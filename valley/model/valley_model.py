"""Same import path as the reference's valley/model/valley_model.py; implementation in valley_amd."""
from valley_amd.valley_model import (ValleyConfig, ValleyLlamaForCausalLM, ValleyLlamaModel,  # noqa: F401
                                     build_vision_tower)
