from .train import DiscriminatorTrainer, GeneratorTrainer, training_loop
