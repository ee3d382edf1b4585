"""CPU oracle: test infrastructure only (see qk_oracle.py). Never imported by quake_amd."""
