"""Defaults of the reference's module-level constants that reach the hot path (config.py:8-36
of the reference).  The operators take them as explicit arguments; these are only defaults."""
SAMPLE_NUM = 128            # config.py:8
CD_W1 = 1.0                 # config.py:11
CD_W2 = 1.0                 # config.py:12
MANUAL_SEED = 1234          # config.py:20
VP_CLAMP_MIN = 0.01         # config.py:22
VP_CLAMP_MAX = 0.8          # config.py:23
IS_SIGMOID = True           # config.py:25
VOLUME_RESTRICT = [8, 10, 10]   # config.py:26
SILHOUETTE_LOSS_FUNC = 'L1'  # config.py:27
CUBOID_NUM = 0              # config.py:33
SPHERE_NUM = 16             # config.py:34
CONE_NUM = 0                # config.py:35
VP_NUM = CUBOID_NUM + SPHERE_NUM + CONE_NUM
IMG_SIZE = 128              # config.py:49

# soft raster (new operator; specification: oracle/vpn_oracle.py::raster)
RASTER_SIGMA = 0.05
RASTER_GAMMA = 0.1
RASTER_Z_FAR = 2.0
MESH_RASTER_SIGMA = 1e-4    # triangle-mesh silhouettes (oracle/vpn_oracle.py::mesh_raster): NDC^2, ~1 pixel of softness at 128^2
