from gmlm_oracle import degree  # noqa: F401
