from .init import weights_init
from .experiment import Experiment
from .melgan import MultiScaleMelGanExperiment
from .realmelgan import RealMelGanExperiment
from .featureexperiment import TwoDimGeneratorFeatureExperiment
