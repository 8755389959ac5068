from gmlm_oracle import OracleRGCNConv as RGCNConv, OracleGraphNorm as GraphNorm  # noqa: F401
