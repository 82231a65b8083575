def parallel_wrapper_fn(env_fn):   # imported by harlsustaindc_env.py:9, never called on the recorded path
    raise NotImplementedError
