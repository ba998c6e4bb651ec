from .full import MelGanGenerator
