"""CPU oracle of the render path -- test infrastructure only (see nerf_oracle.py header)."""
