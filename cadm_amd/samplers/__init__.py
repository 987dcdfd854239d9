from .model_sample_processor import ModelSampleProcessor  # noqa: F401
