"""Drop-in import path: `from valley.model.valley_model import ValleyLlamaForCausalLM` resolves to the
MI355X implementation (valley_amd).  Only the hot-path modules of the reference exist here."""
