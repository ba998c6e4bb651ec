from .init import weights_init
