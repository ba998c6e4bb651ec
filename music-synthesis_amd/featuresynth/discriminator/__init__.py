from .full import FullDiscriminator
from .melgan import MelGanDiscriminator
