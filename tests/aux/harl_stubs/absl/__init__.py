"""Stand-in for the third-party `absl` package (absent from this image): harl/envs/__init__.py parses an empty flag list with it."""
