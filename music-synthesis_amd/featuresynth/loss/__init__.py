from .loss import (hinge_discriminator_loss, hinge_generator_loss, least_squares_disc_loss,
                   least_squares_generator_loss, mel_gan_disc_loss, mel_gan_feature_loss,
                   mel_gan_gen_loss)
